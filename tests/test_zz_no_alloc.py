"""GPU test of the entry point that allocates nothing (tvbf_cosine_topk_static_f32): called through
the C ABI, its lists must equal tvbf_cosine_topk_f32's bit for bit (integer indices and counts, and
the fp32 scores are the same fixed-order sums) on two calls in a row, the second on a smaller
input -- the static scratch keeps the first call's contents, and nothing of them may leak."""

from __future__ import annotations

import pytest

pytestmark = pytest.mark.gpu


def _outputs(torch, nq, k, dev):
    # poisoned, so that slots the call does not write show up
    return (torch.full((nq, k), -7, dtype=torch.int32, device=dev),
            torch.full((nq, k), -7.0, dtype=torch.float32, device=dev),
            torch.full((nq,), -7, dtype=torch.int32, device=dev))


def _both(lib, torch, z, nq, q_row0, n, k, min_sim, dev):
    """(static entry point's outputs, workspace entry point's outputs) for the same arguments."""
    from tvbingefriend_recommendation_service_amd import native

    ld = z.shape[1]
    stream = torch.cuda.current_stream().cuda_stream
    zq = z.data_ptr() + q_row0 * ld * 4
    a = _outputs(torch, nq, k, dev)
    rc = lib.tvbf_cosine_topk_static_f32(zq, nq, q_row0, z.data_ptr(), n, ld, ld, k, min_sim, 1,
                                         a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), stream)
    native.check(rc, "tvbf_cosine_topk_static_f32")
    need = int(lib.tvbf_topk_workspace_bytes(nq, n, ld, k))
    ws = torch.empty((need + 15) // 16 * 2, dtype=torch.int64, device=dev)
    b = _outputs(torch, nq, k, dev)
    rc = lib.tvbf_cosine_topk_f32(zq, nq, q_row0, z.data_ptr(), n, ld, ld, k, min_sim, 1,
                                  b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), ws.data_ptr(),
                                  ws.numel() * 8, stream)
    native.check(rc, "tvbf_cosine_topk_f32")
    torch.cuda.synchronize()
    return a, b


def test_static_scratch_entry_point_equals_workspace_entry_point(pkg, device):
    import torch

    from tvbingefriend_recommendation_service_amd import engine, native

    lib = native.load()
    torch.manual_seed(4242)
    # first call: 5000 shows x 256 columns (LDS-DMA loop, several list slots per row), all rows, K=20
    x1 = torch.randn((5000, 256), device=device)
    x1[:, :32] = (x1[:, :32] > 1.2).float()  # a discrete block: exact ties among the scores
    z1 = engine.normalize_blocks(x1, [0, 32, 256], [0.6, 0.8])
    a, b = _both(lib, torch, z1, 5000, 0, 5000, 20, 0.05, device)
    for got, want, name in zip(a, b, ("idx", "score", "count")):
        assert torch.equal(got, want), f"first call: {name} differs"
    assert int(a[2].sum()) > 0
    # second call, smaller in every size: a shard's rows (not from row 0) of 1337 shows x 96 columns
    # (register-staged loop), K=7, another threshold; stale lists of the first call must not matter
    x2 = torch.randn((1337, 96), device=device)
    z2 = engine.normalize_blocks(x2, [0, 96], [1.0])
    a, b = _both(lib, torch, z2, 300, 640, 1337, 7, -1.0, device)
    for got, want, name in zip(a, b, ("idx", "score", "count")):
        assert torch.equal(got, want), f"second call: {name} differs"
    assert int(a[2].min()) == 7
    # and the first input again: the scratch now holds the smaller call's contents
    a, b = _both(lib, torch, z1, 5000, 0, 5000, 20, 0.05, device)
    for got, want, name in zip(a, b, ("idx", "score", "count")):
        assert torch.equal(got, want), f"third call: {name} differs"


def test_static_scratch_entry_point_rejects_sizes_beyond_its_buffer(pkg, device):
    import torch

    from tvbingefriend_recommendation_service_amd import engine, native

    lib = native.load()
    n, k = 60000, 100  # 60000 rows x 100 keys x 8 bytes per list slot alone exceed 32 MiB
    assert int(lib.tvbf_topk_workspace_bytes(n, n, 64, k)) > native.TVBF_STATIC_SCRATCH_BYTES
    z = torch.zeros((n, 64), device=device)
    o = _outputs(torch, n, k, device)
    rc = lib.tvbf_cosine_topk_static_f32(z.data_ptr(), n, 0, z.data_ptr(), n, 64, 64, k, 0.0, 1, o[0].data_ptr(),
                                         o[1].data_ptr(), o[2].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == -2
    assert b"static buffer" in lib.tvbf_last_error()
    torch.cuda.synchronize()
    assert int(o[2][0]) == -7  # nothing ran


def test_static_scratch_refuses_a_second_stream_while_the_first_is_busy(pkg, device):
    """All calls on a GPU share the library's scratch array: a call on another stream than the
    previous call's is refused (TVBF_E_BUSY) while that stream still has work in flight, and served
    once it has drained -- with the same lists."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine, native

    lib = native.load()
    z = engine.normalize_blocks(torch.randn((3000, 128), device=device), [0, 128], [1.0])
    nq = n = 3000
    k, ld = 10, z.shape[1]
    s1, s2 = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    big = torch.randn((8192, 8192), device=device)
    a, b = _outputs(torch, nq, k, device), _outputs(torch, nq, k, device)
    torch.cuda.synchronize()  # the stream of the earlier calls has drained

    def call(stream, outs):
        return lib.tvbf_cosine_topk_static_f32(z.data_ptr(), nq, 0, z.data_ptr(), n, ld, ld, k, -1.0, 1,
                                               outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), stream.cuda_stream)

    with torch.cuda.stream(s1):
        assert call(s1, a) == 0  # everything idle (the earlier calls' stream has drained): the scratch goes to s1
        for _ in range(20):  # and s1 stays busy for tens of milliseconds
            big = big @ big * 1e-4
    rc = call(s2, b)
    assert rc == -4, f"expected TVBF_E_BUSY while the first stream is busy, got {rc}"
    assert b"another stream" in lib.tvbf_last_error()
    s1.synchronize()
    assert call(s2, b) == 0  # the first stream has drained: the scratch changes hands
    torch.cuda.synchronize()
    for got, want, name in zip(b, a, ("idx", "score", "count")):
        assert torch.equal(got, want), name
