"""Device-side assembly of the feature blocks (dense + CSR sources -> normalised Z)."""

from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import similarity_oracle as so
from synth import hybrid_like_blocks

pytestmark = pytest.mark.gpu


def _device_sources(engine, torch, device, sources):
    out = []
    for s in sources:
        if s[0] == "dense":
            out.append(engine.DenseSource(torch.from_numpy(s[1]).to(device), s[2]))
        else:
            _, indptr, indices, data, width, zc = s
            out.append(engine.CsrSource(torch.from_numpy(indptr).to(device), torch.from_numpy(indices).to(device),
                                        torch.from_numpy(data).to(device), width, zc))
    return out


@pytest.mark.parametrize("variant", ["computer", "populate"])
@pytest.mark.parametrize("n,d_text", [(1001, 960), (257, 61)])
def test_assembly_equals_packed_route_bitwise(pkg, device, variant, n, d_text):
    """Stored arrays (int64 / CSR float64 / float64 / bool) assembled on the device give the
    same bits as the host-packed fp32 matrix through the plain normalise kernel."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine, features as feat

    f = hybrid_like_blocks(n, 77 + n, d_text=d_text, as_features=True)
    x, layout = feat.pack_features(f, 0.4, 0.5, 0.1, variant)
    z_ref = engine.normalize_blocks(torch.from_numpy(x).to(device), layout.offsets, layout.scales())
    sources, layout2 = feat.block_sources(f, 0.4, 0.5, 0.1, variant)
    assert layout2.offsets == layout.offsets and layout2.weights == layout.weights
    z = engine.assemble_normalize(_device_sources(engine, torch, device, sources), n, layout2.offsets,
                                  layout2.scales())
    torch.cuda.synchronize()
    assert z.shape == z_ref.shape
    assert torch.equal(z, z_ref)
    # and both agree with the float64 oracle normalisation
    blocks = [x[:, layout.offsets[b]:layout.offsets[b + 1]].astype(np.float64) for b in range(layout.nb)]
    ref = np.hstack([np.sqrt(w) * so.l2_normalize_rows(b) for b, w in zip(blocks, layout.weights)])
    assert np.abs(z.cpu().numpy()[:, :layout.d] - ref).max() <= 2e-7


def test_csr_variants_duplicates_and_wide_rows(pkg, device):
    """int64 indices, fp32 data, repeated entries, out-of-range columns, an empty matrix and a row
    too wide for more than one row per workgroup."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    rng = np.random.default_rng(5)
    n, width = 300, 13001
    dense = np.zeros((n, width), dtype=np.float32)
    rows, cols, vals = [], [], []
    for i in range(n):
        if i % 17 == 0:
            continue  # empty rows
        c = rng.choice(width, size=int(rng.integers(1, 90)), replace=False)
        v = rng.standard_normal(c.size).astype(np.float32)
        # every third entry is split in two halves that must add up again
        for cc, vv in zip(c, v):
            if cc % 3 == 0:
                rows += [i, i]; cols += [cc, cc]; vals += [vv * 0.25, vv * 0.75]
                dense[i, cc] = np.float32(vv * 0.25) + np.float32(vv * 0.75)
            else:
                rows.append(i); cols.append(cc); vals.append(vv)
                dense[i, cc] = vv
    order = np.lexsort((cols, rows))
    rows, cols, vals = np.array(rows)[order], np.array(cols)[order], np.array(vals, dtype=np.float32)[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr)
    # a stray entry outside the block in the last row is ignored
    cols = np.append(cols, width + 5).astype(np.int64)
    vals = np.append(vals, np.float32(9.0))
    indptr[-1] += 1
    side = rng.standard_normal((n, 6))
    d = width + 3 + 8  # csr block padded to a multiple of 4, then the dense block
    off = [0, width + 3, d]
    src = [engine.CsrSource(torch.from_numpy(indptr).to(device), torch.from_numpy(cols).to(device),
                            torch.from_numpy(vals).to(device), width, 0),
           engine.DenseSource(torch.from_numpy(side).to(device), width + 3)]
    z = engine.assemble_normalize(src, n, off, [1.0, 0.5])
    torch.cuda.synchronize()
    zz = z.cpu().numpy()
    ref_a = so.l2_normalize_rows(dense.astype(np.float64))
    ref_b = 0.5 * so.l2_normalize_rows(side)
    assert np.abs(zz[:, :width] - ref_a).max() <= 3e-7
    assert np.abs(zz[:, width + 3:width + 9] - ref_b).max() <= 3e-7
    assert np.all(zz[:, width:width + 3] == 0) and np.all(zz[:, width + 9:] == 0)
    assert np.all(zz[::17, :width] == 0)

    # a matrix without entries: null indices / data pointers are accepted
    empty = sp.csr_matrix((4, 8), dtype=np.float64)
    src = [engine.CsrSource(torch.from_numpy(empty.indptr).to(device), torch.from_numpy(empty.indices).to(device),
                            torch.from_numpy(empty.data).to(device), 8, 0)]
    z = engine.assemble_normalize(src, 4, [0, 8], [1.0])
    torch.cuda.synchronize()
    assert z.shape[0] == 4 and float(z.abs().max()) == 0.0


def test_assemble_rejects_bad_arguments(pkg, device):
    import torch

    from tvbingefriend_recommendation_service_amd import engine, native

    x = torch.zeros((4, 8), dtype=torch.float64, device=device)
    with pytest.raises(native.NativeLibraryError, match="leave"):
        engine.assemble_normalize([engine.DenseSource(x, 4)], 4, [0, 8], [1.0])
    with pytest.raises(TypeError):
        engine.assemble_normalize([engine.DenseSource(x.to(torch.float16), 0)], 4, [0, 8], [1.0])
    with pytest.raises(ValueError):
        engine.assemble_normalize([engine.DenseSource(x, 0)], 5, [0, 8], [1.0])
