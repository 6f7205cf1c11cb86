"""Column-side load of the symmetric sweep with one and two levels on the small tie-rich test
matrices (usage: ring_probe.py N SHARDS)."""
import os, sys
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
from pathlib import Path
root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np, torch
import tvbf_amd  # noqa
from tvbingefriend_recommendation_service_amd import engine
from synth import hybrid_like_blocks

n, shards = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g, t, p, ty, la = hybrid_like_blocks(n, seed=900 + n + 2, d_genre=16, d_text=208)
x = np.ascontiguousarray(np.hstack([g, t, p, ty, la]), dtype=np.float32)
z = engine.normalize_blocks(torch.from_numpy(x).to(dev), [0, 16, 224, x.shape[1]], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)])
os.environ["TVBF_SYM_SHARDS"] = str(shards)
for k, ms in ((20, 0.1), (31, 0.05)):
    for ring in (1, 2, 3, 4):
        os.environ["TVBF_SYM_RING"] = str(ring)
        nbytes = engine.symmetric_workspace_bytes(n, z.shape[1], k)
        ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
        top = engine.cosine_topk_symmetric(z, k, ms, workspace=ws)
        print(f"k={k} ms={ms} ring={ring}:", engine.symmetric_inspect(ws, n, z.shape[1], k), flush=True)
