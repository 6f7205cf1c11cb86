"""Achievable HBM bandwidth of plain torch fills, copies and reductions on this GPU (the yardstick for
the memory-bound helper kernels).  Usage: python tools/bw_probe.py"""
import torch
dev=torch.device("cuda:0")
def timed(fn,reps=20):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps
for mb in (410, 1640):
    n=mb*1000*1000//4
    z=torch.empty(n,dtype=torch.float32,device=dev); x=torch.randn(n,dtype=torch.float32,device=dev)
    t=timed(lambda: z.zero_()); print(mb,"MB zero_", round(t,4),"ms", round(mb/1e3/(t*1e-3),1),"GB/s write")
    t=timed(lambda: z.copy_(x)); print(mb,"MB copy_", round(t,4),"ms", round(2*mb/1e3/(t*1e-3),1),"GB/s r+w")
    t=timed(lambda: x.sum()); print(mb,"MB sum", round(t,4),"ms", round(mb/1e3/(t*1e-3),1),"GB/s read")
    t=timed(lambda: torch.mul(x,2.0,out=z)); print(mb,"MB mul", round(t,4),"ms", round(2*mb/1e3/(t*1e-3),1),"GB/s r+w")
