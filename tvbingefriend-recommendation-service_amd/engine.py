"""Device-side operations of the similarity path on torch tensors.

torch is plumbing here: it owns the HBM buffers and the HIP stream; all
arithmetic happens in the kernels of csrc/ reached through the C ABI
(:mod:`native`).  Every function launches on torch's current stream and
returns without synchronising.
"""

from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass

import torch

from . import native

LD_ALIGN = 32  # Z rows are padded to a multiple of one k-tile (32 floats = 128 B)


def _require_cuda(t: torch.Tensor, name: str, dtype=torch.float32) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise native.NativeLibraryError(f"{name} must be a tensor on the GPU (there is no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def padded_ld(d: int) -> int:
    """Row length of Z for d feature columns: a multiple of two k-tiles (64 floats), which the
    LDS-DMA tile loop, the symmetric sweeps, the stored-matrix route and the one-sweep statistics
    need (zero columns change no score; the extra products cost less than those routes save --
    the statistics of a 32-column block of 100,000 shows take 9.8 ms at 64 floats per row against
    42 ms at 32)."""
    d = max(int(d), 1)
    return (d + 2 * LD_ALIGN - 1) // (2 * LD_ALIGN) * (2 * LD_ALIGN)


def normalize_blocks(x: torch.Tensor, offsets, scales, out: torch.Tensor | None = None) -> torch.Tensor:
    """Z[:, blk] = scale_blk * x[:, blk] / ||x[:, blk]|| per row (zero blocks stay zero).

    ``x`` is (n, d) fp32 on the GPU with unit column stride.  Returns (n, ld)
    with ld = d rounded up to 32 and zero padding columns.
    """
    _require_cuda(x, "x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be 2-D with contiguous rows")
    n, d = x.shape
    if int(offsets[-1]) != d:
        raise ValueError("last block offset must equal the number of columns")
    ld = padded_ld(d)
    if out is None:
        out = torch.empty((n, ld), dtype=torch.float32, device=x.device)
    _require_cuda(out, "out")
    if out.shape != (n, ld) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous ({n}, {ld}) tensor")
    off, sc, nb = native.block_arrays(offsets, scales)
    lib = native.load()
    with torch.cuda.device(x.device):
        rc = lib.tvbf_normalize_blocks_f32(x.data_ptr(), n, d, x.stride(0), off, sc, nb,
                                           out.data_ptr(), ld, _stream_ptr())
    native.check(rc, "tvbf_normalize_blocks_f32")
    return out


@dataclass
class DenseSource:
    """A dense (n, width) block on the device: fp32, fp64, int64, bool or uint8."""

    data: torch.Tensor
    z_col: int


@dataclass
class CsrSource:
    """A CSR block on the device: indptr (n+1), indices (nnz) both int32 or both int64,
    data (nnz) fp32 or fp64; ``width`` columns."""

    indptr: torch.Tensor
    indices: torch.Tensor
    data: torch.Tensor
    width: int
    z_col: int


_DENSE_KINDS = {torch.float32: native.SRC_DENSE_F32, torch.float64: native.SRC_DENSE_F64,
                torch.int64: native.SRC_DENSE_I64, torch.bool: native.SRC_DENSE_U8,
                torch.uint8: native.SRC_DENSE_U8}


def upload_sources(sources, device) -> list:
    """The host-side sources of :func:`features.block_sources` (("dense", array, z_col) /
    ("csr", indptr, indices, data, width, z_col)) as device sources, arrays uploaded as stored."""
    out = []
    for src in sources:
        if src[0] == "dense":
            out.append(DenseSource(torch.from_numpy(src[1]).to(device), src[2]))
        else:
            _, indptr, indices, data, width, z_col = src
            out.append(CsrSource(torch.from_numpy(indptr).to(device), torch.from_numpy(indices).to(device),
                                 torch.from_numpy(data).to(device), width, z_col))
    return out


def slice_sources(sources, row0: int, rows: int) -> list:
    """The device sources restricted to shows row0 .. row0 + rows (views, nothing copied): dense
    blocks by rows, CSR blocks by their slice of indptr (its entries keep pointing into the whole
    indices / data arrays)."""
    out = []
    for s in sources:
        if isinstance(s, DenseSource):
            out.append(DenseSource(s.data[row0:row0 + rows], s.z_col))
        else:
            out.append(CsrSource(s.indptr[row0:row0 + rows + 1], s.indices, s.data, s.width, s.z_col))
    return out


def sources_rows(sources) -> int:
    """Number of shows the device sources describe."""
    s = sources[0]
    return int(s.data.shape[0]) if isinstance(s, DenseSource) else int(s.indptr.numel()) - 1


def _source_array(sources, n: int):
    """ctypes array of struct tvbf_source for device sources of n rows, and their device."""
    if not 1 <= len(sources) <= native.TVBF_MAX_SOURCES:
        raise ValueError(f"1..{native.TVBF_MAX_SOURCES} sources are supported")
    arr = (native.Source * len(sources))()
    device = None
    for i, s in enumerate(sources):
        if isinstance(s, DenseSource):
            t = s.data
            _require_cuda(t, "dense source", None)
            if t.dim() != 2 or t.shape[0] != n or (t.shape[1] > 1 and t.stride(1) != 1):
                raise ValueError("dense sources must be (n, width) with contiguous rows")
            if t.dtype not in _DENSE_KINDS:
                raise TypeError(f"unsupported dense source dtype {t.dtype}")
            arr[i] = native.Source(_DENSE_KINDS[t.dtype], 0, t.shape[1], int(s.z_col), t.data_ptr() or None,
                                   t.stride(0) if t.shape[0] > 1 else t.shape[1], None, None)
            device = t.device
        elif isinstance(s, CsrSource):
            for t, name in ((s.indptr, "indptr"), (s.indices, "indices"), (s.data, "data")):
                _require_cuda(t, name, None)
                if t.dim() != 1 or not t.is_contiguous():
                    raise ValueError(f"CSR {name} must be a contiguous vector")
            if s.indptr.numel() != n + 1 or s.indices.numel() != s.data.numel():
                raise ValueError("CSR arrays have inconsistent lengths")
            if s.indptr.dtype != s.indices.dtype or s.indptr.dtype not in (torch.int32, torch.int64):
                raise TypeError("CSR indptr / indices must both be int32 or both int64")
            if s.data.dtype not in (torch.float32, torch.float64):
                raise TypeError("CSR data must be fp32 or fp64")
            kind = native.SRC_CSR_F32 if s.data.dtype == torch.float32 else native.SRC_CSR_F64
            arr[i] = native.Source(kind, int(s.indptr.dtype == torch.int64), int(s.width), int(s.z_col),
                                   s.data.data_ptr() or None, 0, s.indptr.data_ptr(), s.indices.data_ptr() or None)
            device = s.indptr.device
        else:
            raise TypeError("sources must be DenseSource or CsrSource")
    return arr, device


def assemble_normalize(sources, n: int, offsets, scales, out: torch.Tensor | None = None) -> torch.Tensor:
    """Z (n, ld) from feature blocks that are already on the device in their stored form
    (dense arrays and CSR triples), normalised and weighted block-wise like
    :func:`normalize_blocks`; no dense copy of a sparse block is ever made."""
    d = int(offsets[-1])
    ld = padded_ld(d)
    arr, device = _source_array(sources, n)
    if out is None:
        out = torch.empty((n, ld), dtype=torch.float32, device=device)
    _require_cuda(out, "out")
    if out.shape != (n, ld) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous ({n}, {ld}) tensor")
    off, sc, nb = native.block_arrays(offsets, scales)
    lib = native.load()
    with torch.cuda.device(out.device):
        rc = lib.tvbf_assemble_normalize_f32(arr, len(sources), n, d, off, sc, nb, out.data_ptr(), ld, _stream_ptr())
    native.check(rc, "tvbf_assemble_normalize_f32")
    return out


@dataclass
class TopK:
    """Per-row neighbours: idx (nq,k) int32 (-1 = unused), score (nq,k) fp32,
    count (nq,) int32; rows ordered by score descending, then index ascending."""

    idx: torch.Tensor
    score: torch.Tensor
    count: torch.Tensor


def topk_workspace_bytes(nq: int, n: int, d: int, k: int, device=None) -> int:
    lib = native.load()
    if device is not None and torch.cuda.is_available():
        with torch.cuda.device(device):
            return int(lib.tvbf_topk_workspace_bytes(nq, n, d, k))
    return int(lib.tvbf_topk_workspace_bytes(nq, n, d, k))


def cosine_topk(z_all: torch.Tensor, k: int, min_similarity: float, *, q_row0: int = 0,
                nq: int | None = None, n: int | None = None, exclude_self: bool = True,
                z_q: torch.Tensor | None = None, workspace: torch.Tensor | None = None,
                out: TopK | None = None) -> TopK:
    """Top-k of (z_q @ z_all.T) per query row without materialising the scores.

    ``z_all`` is (rows >= n, ld) fp32; the query rows default to rows
    ``q_row0 .. q_row0 + nq`` of ``z_all`` (pass ``z_q`` for a separate block).
    """
    _require_cuda(z_all, "z_all")
    if z_all.dim() != 2 or not z_all.is_contiguous():
        raise ValueError("z_all must be a contiguous 2-D tensor")
    n = z_all.shape[0] if n is None else int(n)
    ld = z_all.shape[1]
    if z_q is None:
        nq = n - q_row0 if nq is None else int(nq)
        if q_row0 < 0 or q_row0 + nq > z_all.shape[0]:
            raise ValueError("query rows outside z_all")
        zq_ptr = z_all.data_ptr() + q_row0 * ld * 4
    else:
        _require_cuda(z_q, "z_q")
        if z_q.dim() != 2 or not z_q.is_contiguous() or z_q.shape[1] != ld:
            raise ValueError("z_q must be contiguous with the same row length as z_all")
        nq = z_q.shape[0] if nq is None else int(nq)
        zq_ptr = z_q.data_ptr()
    if not 1 <= n <= z_all.shape[0]:
        raise ValueError("n outside z_all")
    if not 1 <= k <= native.TVBF_MAX_K:
        raise ValueError(f"k must be in 1..{native.TVBF_MAX_K}")
    dev = z_all.device
    if out is None:
        out = TopK(torch.empty((nq, k), dtype=torch.int32, device=dev),
                   torch.empty((nq, k), dtype=torch.float32, device=dev),
                   torch.empty((nq,), dtype=torch.int32, device=dev))
    lib = native.load()
    with torch.cuda.device(dev):
        need = int(lib.tvbf_topk_workspace_bytes(max(nq, 1), n, ld, k))
        if workspace is None:
            workspace = torch.empty((need + 15) // 16 * 2, dtype=torch.int64, device=dev)
        ws_bytes = workspace.numel() * workspace.element_size()
        rc = lib.tvbf_cosine_topk_f32(zq_ptr, nq, q_row0, z_all.data_ptr(), n, ld, ld, k,
                                      float(min_similarity), 1 if exclude_self else 0,
                                      out.idx.data_ptr(), out.score.data_ptr(), out.count.data_ptr(),
                                      workspace.data_ptr(), ws_bytes, _stream_ptr())
    native.check(rc, "tvbf_cosine_topk_f32")
    return out


def symmetric_workspace_bytes(n: int, ld: int, k: int) -> int:
    """Workspace of :func:`cosine_topk_symmetric`; 0 when that route does not take the shape
    (row length not a multiple of 64, fewer than 16 blocks of 128 rows)."""
    return int(native.load().tvbf_cosine_topk_sym_workspace_bytes(int(n), int(ld), int(k)))


def symmetric_tiles(n: int, ld: int, k: int, device=None) -> int:
    """128 x 128 score tiles that :func:`cosine_topk_symmetric` computes for these sizes on ``device``
    (0: shape not taken); a plain sweep computes ceil(n / 128) ** 2."""
    lib = native.load()
    if device is not None:
        with torch.cuda.device(device):
            return int(lib.tvbf_cosine_topk_sym_tiles(int(n), int(ld), int(k)))
    return int(lib.tvbf_cosine_topk_sym_tiles(int(n), int(ld), int(k)))


def cosine_topk_symmetric(z: torch.Tensor, k: int, min_similarity: float, *, n: int | None = None,
                          exclude_self: bool = True, workspace: torch.Tensor | None = None,
                          out: TopK | None = None) -> TopK:
    """Top-k of every row of ``z[:n] @ z[:n].T`` computing each pair of shows once: the score
    matrix is symmetric bit for bit, so the lists equal :func:`cosine_topk`'s exactly."""
    _require_cuda(z, "z")
    if z.dim() != 2 or not z.is_contiguous():
        raise ValueError("z must be a contiguous 2-D tensor")
    n = z.shape[0] if n is None else int(n)
    if not 1 <= n <= z.shape[0]:
        raise ValueError("n outside z")
    ld = z.shape[1]
    dev = z.device
    lib = native.load()
    with torch.cuda.device(dev):
        need = int(lib.tvbf_cosine_topk_sym_workspace_bytes(n, ld, k))
        if need == 0:
            raise ValueError("the symmetric route does not take this shape (see symmetric_workspace_bytes)")
        if workspace is None:
            workspace = torch.empty((need + 15) // 16 * 2, dtype=torch.int64, device=dev)
        if out is None:
            out = TopK(torch.empty((n, k), dtype=torch.int32, device=dev),
                       torch.empty((n, k), dtype=torch.float32, device=dev),
                       torch.empty((n,), dtype=torch.int32, device=dev))
        rc = lib.tvbf_cosine_topk_sym_f32(z.data_ptr(), n, ld, ld, k, float(min_similarity),
                                          1 if exclude_self else 0, out.idx.data_ptr(), out.score.data_ptr(),
                                          out.count.data_ptr(), workspace.data_ptr(),
                                          workspace.numel() * workspace.element_size(), _stream_ptr())
    native.check(rc, "tvbf_cosine_topk_sym_f32")
    return out


OWN_MATRIX_MAX_ROWS = 32768  # blocks up to this size take the stored-matrix route when a buffer is given


def own_matrix_bytes(rows: int, ld: int) -> int:
    """Bytes of the matrix buffer :meth:`TopkSession.accumulate_own_matrix` needs for a block of
    ``rows`` rows; 0 when that route does not take the shape (row length not a multiple of 64,
    more than 32,768 rows: the fused sweeps are the better choice there)."""
    if rows < 1 or rows > OWN_MATRIX_MAX_ROWS:
        return 0
    return int(native.load().tvbf_topk_matrix_bytes(int(rows), int(ld)))


def symmetric_inspect(workspace: torch.Tensor, n: int, ld: int, k: int) -> dict:
    """What the column side of the last :func:`cosine_topk_symmetric` call on ``workspace`` did
    (synchronises): overflow flag, candidates appended, the busiest show."""
    import ctypes

    out = (ctypes.c_int64 * 4)()
    with torch.cuda.device(workspace.device):
        rc = native.load().tvbf_cosine_topk_sym_inspect(workspace.data_ptr(), int(n), int(ld), int(k), out, _stream_ptr())
    native.check(rc, "tvbf_cosine_topk_sym_inspect")
    return {"overflowed": int(out[0]), "appended": int(out[1]), "max_per_show": int(out[2]), "busiest_show": int(out[3])}


MEMORY_HEADROOM = 0.9  # share of the currently available device memory a workspace may take


def available_bytes(device) -> int:
    """Device memory a new torch allocation can use right now: what the driver reports free plus
    what torch's caching allocator holds without using it."""
    free, _ = torch.cuda.mem_get_info(device)
    return int(free) + int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device))


def choose_catalogue_route(symmetric_bytes: int, plain_bytes: int, available: int, choice: str = "auto") -> str:
    """Which whole-catalogue call to make: "symmetric" (every pair once; its workspace holds score
    matrices of up to 4 n^2 bytes for a small catalogue, the shards' own blocks -- up to 8 GiB -- and
    8 KB of column-side buffers per show for a large one) while that workspace fits into
    ``MEMORY_HEADROOM`` of the available memory, else "plain" (tens of MB).  Both give the same
    lists bit for bit, so falling back costs time only.  ``choice`` "plain" / "symmetric" forces a
    route (TVBF_SWEEP)."""
    if choice == "plain" or symmetric_bytes <= 0:
        return "plain"
    if choice == "symmetric" or symmetric_bytes <= MEMORY_HEADROOM * available:
        return "symmetric"
    if plain_bytes > MEMORY_HEADROOM * available:
        raise MemoryError(f"the top-K sweep needs {plain_bytes} bytes of workspace, {available} are available")
    return "plain"


def catalogue_topk(z: torch.Tensor, k: int, min_similarity: float, *, n: int | None = None,
                   exclude_self: bool = True) -> TopK:
    """Top-k of every show against the whole catalogue: the symmetric call when the library takes
    the shape (row length a multiple of 64; it stores the scores of a small catalogue as a matrix
    and runs the fused symmetric sweeps on a large one) and its workspace fits into the memory that
    is available, the plain sweep otherwise.  All routes give the same lists bit for bit.
    TVBF_SWEEP=plain|symmetric forces one."""
    import os

    n = z.shape[0] if n is None else int(n)
    with torch.cuda.device(z.device):
        sym = symmetric_workspace_bytes(n, z.shape[1], k)
        plain = int(native.load().tvbf_topk_workspace_bytes(n, n, z.shape[1], k))
    route = choose_catalogue_route(sym, plain, available_bytes(z.device), os.environ.get("TVBF_SWEEP", "auto"))
    if route == "symmetric":
        return cosine_topk_symmetric(z, k, min_similarity, n=n, exclude_self=exclude_self)
    return cosine_topk(z, k, min_similarity, n=n, exclude_self=exclude_self)


def component_scores(z_all: torch.Tensor, offsets, scales, idx: torch.Tensor, *, q_row0: int = 0,
                     n: int | None = None, z_q: torch.Tensor | None = None) -> torch.Tensor:
    """Per-block cosines (nq, k, nb) of the kept pairs listed in ``idx``."""
    _require_cuda(z_all, "z_all")
    _require_cuda(idx, "idx", torch.int32)
    if not idx.is_contiguous() or idx.dim() != 2:
        raise ValueError("idx must be a contiguous (nq, k) tensor")
    nq, k = idx.shape
    n = z_all.shape[0] if n is None else int(n)
    ld = z_all.shape[1]
    if z_q is None:
        if q_row0 < 0 or q_row0 + nq > z_all.shape[0]:
            raise ValueError("query rows outside z_all")
        zq_ptr = z_all.data_ptr() + q_row0 * ld * 4
    else:
        _require_cuda(z_q, "z_q")
        zq_ptr = z_q.data_ptr()
    off, sc, nb = native.block_arrays(offsets, scales)
    if int(offsets[-1]) > ld:
        raise ValueError("block offsets exceed the row length")
    out = torch.empty((nq, k, nb), dtype=torch.float32, device=z_all.device)
    lib = native.load()
    with torch.cuda.device(z_all.device):
        rc = lib.tvbf_component_scores_f32(zq_ptr, nq, z_all.data_ptr(), n, ld, off, sc, nb,
                                           idx.data_ptr(), k, out.data_ptr(), _stream_ptr())
    native.check(rc, "tvbf_component_scores_f32")
    return out


# how far below min_similarity the fp32 sweep still collects pairs that refine_min_similarity then
# decides in float64; the fp32 error of a hybrid score is about 3e-7 at 1024 columns
THRESHOLD_BAND = 2e-6


def error_margin(ld: int, group_weights=None, base: float = THRESHOLD_BAND) -> float:
    """The allowance for the fp32 error of a score over rows of ``ld`` floats: ``base`` (2e-6, seven
    times the largest error seen over the 10^10 pairs of a 100,000 x 1024 catalogue) up to 1024
    columns, growing with the square root of the row length beyond -- rounding errors of a sum add up
    like a random walk -- and with the sum of the weights' magnitudes beyond 1 (``group_weights``: the
    reference takes any weights, scripts/populate_database.py:190-192 applies them as given, so
    a score -- and its rounding error -- can be as large as that sum; negative weights do not cancel
    errors).  Used as the band below ``min_similarity`` in which the sweeps still collect pairs and
    as the margin of the re-rank's certification."""
    scale = 1.0 if group_weights is None else max(1.0, float(sum(abs(float(w)) for w in group_weights)))
    return float(base) * max(1.0, math.sqrt(int(ld) / 1024.0)) * scale


def refine_min_similarity(top: TopK, comp: torch.Tensor, groups, group_weights, min_similarity: float,
                          band: float = THRESHOLD_BAND) -> None:
    """The reference's ``score < min_similarity`` test re-decided in float64 from fp32 per-block
    cosines (exact when those are exact, e.g. one-hot blocks; :func:`exact_topk` decides from float64
    cosines) for the pairs of ``top`` whose fp32
    score lies within ``band`` of the threshold (in place; ``top`` must come from a sweep run with
    ``min_similarity - band``, ``comp`` (nq, k, nb) are its plain per-block cosines).  ``groups[b]``
    in {0: genre, 1: text, 2: metadata} and the three ``group_weights`` spell the reference's
    expression (wg * g + wt * t) + wm * mean(metadata blocks)."""
    _require_cuda(comp, "comp")
    nq, k = top.idx.shape
    nb = len(groups)
    if comp.shape != (nq, k, nb) or not comp.is_contiguous():
        raise ValueError("comp must be a contiguous (nq, k, len(groups)) tensor")
    if len(group_weights) != 3:
        raise ValueError("three group weights (genre, text, metadata)")
    g = (ctypes.c_int32 * nb)(*[int(v) for v in groups])
    w = (ctypes.c_double * 3)(*[float(v) for v in group_weights])
    with torch.cuda.device(comp.device):
        rc = native.load().tvbf_refine_min_similarity_f64(top.idx.data_ptr(), top.score.data_ptr(), top.count.data_ptr(),
                                                          comp.data_ptr(), nq, k, nb, g, w, float(min_similarity),
                                                          float(band), _stream_ptr())
    native.check(rc, "tvbf_refine_min_similarity_f64")


# ---- float64 re-rank (the reference's order, scores and threshold cut) ----

RERANK_MARGIN = 2e-6  # a float64 score exceeds the fp32 score of the same pair by less than this


@dataclass
class TopK64:
    """Per-row neighbours in the reference's float64 arithmetic: idx (nq,k) int32 (-1 = unused),
    score (nq,k) float64, count (nq,) int32, comp (nq,k,3) float64 genre / text / metadata cosines;
    rows ordered by float64 score descending, equal scores by index ascending.  ``stats`` says how
    many rows needed a longer candidate list (``resweep``) or a scan of all shows (``scan``)."""

    idx: torch.Tensor
    score: torch.Tensor
    count: torch.Tensor
    comp: torch.Tensor
    stats: dict


def candidate_count(k: int) -> int:
    """Candidates per show the fp32 sweep collects for a float64 re-rank to k: k + 8 while the
    lists stay in LDS (k + 8 <= 30), k + 16 beyond (lists in global memory either way)."""
    k = int(k)
    return k + 8 if k + 8 <= 30 else max(k, min(k + 16, native.TVBF_MAX_K))


def threshold_keys(t: torch.Tensor) -> torch.Tensor:
    """The 64-bit bound keys (int64 bit patterns) of fp32 thresholds ``t``: the order-preserving
    integer image of the score in the high word, low word 0 -- every (score, column) key of a score
    >= t is larger.  What a :class:`TopkSession` holds in ``bounds()``."""
    u = t.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    flip = torch.where((u >> 31) != 0, torch.full_like(u, 0xFFFFFFFF), torch.full_like(u, 0x80000000))
    return (u ^ flip) << 32


def resweep_candidate_count(k: int) -> int:
    """Candidates per show of the second stage (rows whose first list ended inside a group of tied
    scores): the sweeps' limit -- the stage starts from a per-row bound, so only the shows inside
    the band of scores that can still matter reach the lists."""
    return max(int(k), native.TVBF_MAX_K)


def raise_if_not_finite(bits: int) -> None:
    """sklearn's input validation messages for the flag word of tvbf_block_norms_f64."""
    if bits & 1:
        raise ValueError("Input contains NaN.")
    if bits & 2:
        raise ValueError("Input contains infinity or a value too large for dtype('float64').")


def block_norms(sources, n: int, offsets, *, check_finite: bool = True, finite_flag: torch.Tensor | None = None) -> torch.Tensor:
    """(n, nb) float64 L2 norms of every show's feature blocks (a zero block counts as 1, as in
    sklearn's normalize).  With ``check_finite`` a NaN or infinite stored value raises ValueError
    with sklearn's message (the reference's cosine_similarity validates its input; synchronises).
    ``finite_flag`` (int32, one zeroed word on the device): the kernel's flag word is left there for
    the caller to read with its next read-back and hand to :func:`raise_if_not_finite` -- no
    synchronisation here."""
    arr, device = _source_array(sources, n)
    nb = len(offsets) - 1
    off = (ctypes.c_int32 * (nb + 1))(*[int(o) for o in offsets])
    norms = torch.empty((n, nb), dtype=torch.float64, device=device)
    flag = finite_flag
    if flag is None and check_finite:
        flag = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        rc = native.load().tvbf_block_norms_f64(arr, len(sources), n, int(offsets[-1]), off, nb, norms.data_ptr(),
                                                flag.data_ptr() if flag is not None else None, _stream_ptr())
    native.check(rc, "tvbf_block_norms_f64")
    if flag is not None and finite_flag is None:
        raise_if_not_finite(int(flag.item()))
    return norms


def rescore_topk(sources, n: int, offsets, groups, group_weights, norms: torch.Tensor, cand: TopK, k: int,
                 min_similarity: float, out: TopK64, *, q_row0: int = 0, row_ids: torch.Tensor | None = None,
                 margin: float = RERANK_MARGIN, exclude_self: bool = True,
                 flagged: torch.Tensor | None = None) -> None:
    """One launch of the float64 re-rank: the candidates ``cand`` (kc per row, from an fp32 sweep)
    of shows ``row_ids`` (int32 on the device; default: shows q_row0 .. q_row0 + rows of cand)
    re-scored from the stored ``sources`` and written, best k first, into ``out`` (whose row 0 is
    show q_row0).  ``flagged`` (int32, 1 + rows, [0] zeroed): receives the shows whose list may miss
    a true neighbour."""
    arr, device = _source_array(sources, n)
    nb = len(offsets) - 1
    nq, kc = cand.idx.shape
    off = (ctypes.c_int32 * (nb + 1))(*[int(o) for o in offsets])
    grp = (ctypes.c_int32 * nb)(*[int(g) for g in groups])
    w = (ctypes.c_double * 3)(*[float(v) for v in group_weights])
    for t, name, dt in ((cand.idx, "cand.idx", torch.int32), (cand.score, "cand.score", torch.float32),
                        (cand.count, "cand.count", torch.int32), (norms, "norms", torch.float64),
                        (out.idx, "out.idx", torch.int32), (out.score, "out.score", torch.float64),
                        (out.comp, "out.comp", torch.float64), (out.count, "out.count", torch.int32)):
        _require_cuda(t, name, dt)
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
    if out.idx.shape[1] != k or out.score.shape != out.idx.shape or out.comp.shape != out.idx.shape + (3,):
        raise ValueError("out must hold (rows, k) idx / score and (rows, k, 3) comp")
    if norms.shape != (n, nb):
        raise ValueError("norms must be (n, nb)")
    if row_ids is not None:
        _require_cuda(row_ids, "row_ids", torch.int32)
        if row_ids.numel() != nq or not row_ids.is_contiguous():
            raise ValueError("row_ids needs one contiguous entry per candidate row")
    elif out.idx.shape[0] < nq:
        raise ValueError("out has fewer rows than cand")
    if flagged is not None:
        _require_cuda(flagged, "flagged", torch.int32)
        if flagged.numel() < 1 + nq:
            raise ValueError("flagged needs 1 + rows entries")
    with torch.cuda.device(device):
        rc = native.load().tvbf_rescore_topk_f64(
            arr, len(sources), n, int(offsets[-1]), off, nb, grp, w, norms.data_ptr(),
            row_ids.data_ptr() if row_ids is not None else None, nq, int(q_row0), cand.idx.data_ptr(),
            cand.score.data_ptr(), cand.count.data_ptr(), kc, int(k), float(min_similarity), float(margin),
            1 if exclude_self else 0, out.idx.data_ptr(), out.score.data_ptr(), out.comp.data_ptr(),
            out.count.data_ptr(), int(q_row0), flagged.data_ptr() if flagged is not None else None, _stream_ptr())
    native.check(rc, "tvbf_rescore_topk_f64")


def score_rows(sources, n: int, offsets, groups, group_weights, norms: torch.Tensor, row_ids: torch.Tensor,
               *, exclude_self: bool = True) -> torch.Tensor:
    """(rows, n) float64 scores of the shows ``row_ids`` (int32 on the device) against all shows,
    in the reference's arithmetic; -inf at a show's own column when ``exclude_self``."""
    arr, device = _source_array(sources, n)
    nb = len(offsets) - 1
    off = (ctypes.c_int32 * (nb + 1))(*[int(o) for o in offsets])
    grp = (ctypes.c_int32 * nb)(*[int(g) for g in groups])
    w = (ctypes.c_double * 3)(*[float(v) for v in group_weights])
    _require_cuda(row_ids, "row_ids", torch.int32)
    h = torch.empty((row_ids.numel(), n), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        rc = native.load().tvbf_score_rows_f64(arr, len(sources), n, int(offsets[-1]), off, nb, grp, w,
                                               norms.data_ptr(), row_ids.data_ptr(), row_ids.numel(),
                                               1 if exclude_self else 0, h.data_ptr(), n, _stream_ptr())
    native.check(rc, "tvbf_score_rows_f64")
    return h


SCAN_BYTES = 1 << 30  # float64 score rows held at a time by the last-resort scan


BAND_LIST_MAX = 1024   # candidates per show the re-rank kernel takes
BAND_BYTES = 512 << 20  # candidate buffers (column + score) of one second-stage launch


def band_candidates(z: torch.Tensor, z_q: torch.Tensor, thresholds: torch.Tensor, cap: int, *, n: int | None = None):
    """Every show whose fp32 score against a query row reaches that ROW'S threshold: (idx (nq, cap)
    int32, score (nq, cap) fp32, count (nq,) int32 -- all such shows, may exceed cap).  ``z_q`` holds
    the query rows (any rows, e.g. gathered ones), ``z`` all n shows; one launch, no lists."""
    _require_cuda(z, "z")
    _require_cuda(z_q, "z_q")
    _require_cuda(thresholds, "thresholds")
    if z.dim() != 2 or not z.is_contiguous() or z_q.dim() != 2 or not z_q.is_contiguous() or z_q.shape[1] != z.shape[1]:
        raise ValueError("z and z_q must be contiguous 2-D tensors of one row length")
    nq = z_q.shape[0]
    n = z.shape[0] if n is None else int(n)
    if thresholds.numel() != nq or not thresholds.is_contiguous():
        raise ValueError("one contiguous fp32 threshold per query row")
    idx = torch.empty((nq, cap), dtype=torch.int32, device=z.device)
    score = torch.empty((nq, cap), dtype=torch.float32, device=z.device)
    count = torch.empty((nq,), dtype=torch.int32, device=z.device)
    with torch.cuda.device(z.device):
        rc = native.load().tvbf_band_candidates_f32(z_q.data_ptr(), nq, z.data_ptr(), n, z.shape[1], z.shape[1],
                                                    thresholds.data_ptr(), int(cap), idx.data_ptr(), score.data_ptr(),
                                                    count.data_ptr(), _stream_ptr())
    native.check(rc, "tvbf_band_candidates_f32")
    return idx, score, count


def _band_stage(z, z_q, sources, n, offsets, groups, group_weights, norms, cand, rows, k, min_similarity, out, q_row0,
                margin, band) -> torch.Tensor:
    """Second stage of :func:`exact_topk` for the shows ``rows`` (int32, sorted): as candidates EVERY
    show whose fp32 score reaches t = max(min_similarity - band, c_K - 2 margin), c_K the K-th best
    fp32 score of the first stage (tvbf_band_candidates_f32: the tile engine's numbers, one score tile
    per workgroup, whatever the number of rows).  A show of the true list has a float64 score of at
    least max(min_similarity, c_K - margin) (K shows are known to score c_K - margin or more), hence an
    fp32 score of at least t: the candidate list is complete, whatever its length, and the re-rank
    of it is final.  Returns the shows with more such candidates than a list takes (groups of a
    thousand tied shows): they go on to the scan."""
    dev = z.device
    left = []
    cap = int(max(k + 1, BAND_LIST_MAX))
    chunk = max(1, BAND_BYTES // (8 * cap))  # shows per launch: their (column, score) buffers stay within BAND_BYTES
    for r0 in range(0, int(rows.numel()), chunk):
        part = rows[r0:r0 + chunk]
        nf = int(part.numel())
        local = part.long() - q_row0
        zq = z.index_select(0, part.long()) if z_q is None else z_q.index_select(0, local)
        floor = torch.full((nf,), min_similarity - band, dtype=torch.float32, device=dev)
        t = torch.maximum(floor, cand.score.index_select(0, local)[:, k - 1] - 2.0 * float(margin)).contiguous()
        idx, score, cnt = band_candidates(z, zq, t, cap, n=n)
        over = cnt >= cap  # more shows in the band than the list takes: not decided here
        count = torch.where(over, torch.zeros_like(cnt), cnt)
        rescore_topk(sources, n, offsets, groups, group_weights, norms, TopK(idx, score, count), k, min_similarity,
                     out, q_row0=q_row0, row_ids=part.contiguous(), margin=margin)
        if bool(over.any().item()):  # rare; the usual case reads one scalar back, no compaction
            left.append(part[over])
    return torch.cat(left).contiguous() if left else rows[:0]


def exact_topk(z: torch.Tensor, sources, offsets, groups, group_weights, k: int, min_similarity: float, *,
               n: int | None = None, q_row0: int = 0, nq: int | None = None, cand: TopK | None = None,
               z_q: torch.Tensor | None = None, resweep=None, resweep_k: int | None = None,
               norms: torch.Tensor | None = None, margin: float | None = None, band: float | None = None,
               check_finite: bool = True) -> TopK64:
    """The reference's top-k lists (scripts/populate_database.py:170-218) for shows
    q_row0 .. q_row0 + nq: float64 scores, float64 order, float64 threshold cut.

    ``z`` is the normalised fp32 matrix of ALL n shows (the sweeps' column operand) and ``sources``
    the stored feature arrays of all n shows on the device.  ``z_q`` (rows q_row0 .. of a query-side
    matrix) is given when the query side differs from ``z`` -- weights applied on one side only.
    Three stages, each only for the rows the previous one could not certify: (1) the fp32 sweep's
    :func:`candidate_count` best (``cand``; computed here when not given) re-ranked in float64;
    (2) every show inside the band of fp32 scores that can still matter (:func:`_band_stage`), re-ranked:
    final unless the band holds more shows than a list takes -- or, with ``resweep(rows, kc)`` /
    ``resweep_k`` given, a longer list from that call or from a plain fp32 sweep of just those rows,
    self not excluded, re-ranked and certified again; (3) their float64 scores against all shows and
    an exact selection.  Reads the number of uncertified rows back between stages."""
    n = z.shape[0] if n is None else int(n)
    nq = n - q_row0 if nq is None else int(nq)
    dev = z.device
    margin = error_margin(z.shape[1], group_weights, RERANK_MARGIN) if margin is None else float(margin)
    band = error_margin(z.shape[1], group_weights) if band is None else float(band)
    # one int32 buffer for everything the host reads back: [0] the finite-check word of the norms
    # kernel, [1] the number of flagged shows, [2:] the flagged shows -- one read-back for both
    words = torch.zeros(2 + nq, dtype=torch.int32, device=dev)
    checked = norms is None and check_finite
    if norms is None:
        norms = block_norms(sources, n, offsets, check_finite=False, finite_flag=words[:1] if checked else None)
    out = TopK64(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float64, device=dev),
                 torch.empty((nq,), dtype=torch.int32, device=dev), torch.empty((nq, k, 3), dtype=torch.float64, device=dev),
                 {"candidates": 0, "resweep": 0, "scan": 0})
    if k > native.TVBF_MAX_K:
        # longer lists than the sweeps keep: every row goes straight to the scan of all shows
        if checked:
            raise_if_not_finite(int(words[0].item()))
        rows = torch.arange(q_row0, q_row0 + nq, dtype=torch.int32, device=dev)
        nf = nq
    else:
        if cand is None:
            kc = candidate_count(k)
            if z_q is not None:
                cand = cosine_topk(z, kc, min_similarity - band, z_q=z_q, q_row0=q_row0, nq=nq, n=n)
            elif q_row0 == 0 and nq == n:
                cand = catalogue_topk(z, kc, min_similarity - band, n=n)
            else:
                cand = cosine_topk(z, kc, min_similarity - band, q_row0=q_row0, nq=nq, n=n)
        kc = cand.idx.shape[1]
        out.stats["candidates"] = kc
        flagged = words[1:]
        rescore_topk(sources, n, offsets, groups, group_weights, norms, cand, k, min_similarity, out, q_row0=q_row0,
                     margin=margin, flagged=flagged)
        bits, nf = words[:2].tolist()  # the one read-back of this stage
        if checked:
            raise_if_not_finite(bits)  # sklearn validates before any arithmetic; nothing was returned yet
        if nf == 0:
            return out
        rows = flagged[1:1 + nf].sort().values.contiguous()
        kc2 = resweep_candidate_count(k) if resweep_k is None else int(resweep_k)
        if resweep is None and resweep_k is None:
            # stage 2: every show inside the band of scores that can still matter as a candidate --
            # a complete list by construction
            out.stats["resweep"] = nf
            rows = _band_stage(z, z_q, sources, n, offsets, groups, group_weights, norms, cand, rows, k, min_similarity,
                               out, q_row0, margin, band)
            nf = int(rows.numel())
            if nf == 0:
                return out
        elif kc2 > kc:  # stage 2: a longer list for just these rows
            out.stats["resweep"] = nf
            if resweep is None:
                def resweep(ids, kk):
                    # A show of the true list has a float64 score of at least max(min_similarity, c_K - margin),
                    # c_K the K-th best fp32 score found (K shows score at least c_K - margin in float64), hence an
                    # fp32 score of at least t = max(min_similarity - band, c_K - 2 margin): the sweep starts with
                    # that as the row's bound, so its lists fill with the few shows of that band only.
                    local = ids.long() - q_row0
                    zq = z.index_select(0, ids.long()) if z_q is None else z_q.index_select(0, local)
                    floor = torch.full((ids.numel(),), min_similarity - band, dtype=torch.float32, device=dev)
                    t = torch.maximum(floor, cand.score.index_select(0, local)[:, k - 1] - 2.0 * float(margin))
                    ses = TopkSession(zq, ids.numel(), 0, kk, min_similarity - band, exclude_self=False)
                    ses.bounds().copy_(threshold_keys(t))
                    ses.accumulate((z, n, 0))
                    return ses.finish()
            cand2 = resweep(rows, kc2)
            flagged2 = torch.zeros(1 + nf, dtype=torch.int32, device=dev)
            rescore_topk(sources, n, offsets, groups, group_weights, norms, cand2, k, min_similarity, out,
                         q_row0=q_row0, row_ids=rows, margin=margin, flagged=flagged2)
            nf = int(flagged2[0].item())
            if nf == 0:
                return out
            rows = flagged2[1:1 + nf].sort().values.contiguous()
    out.stats["scan"] = nf
    batch = max(1, min(nf, SCAN_BYTES // (8 * n), 65535))
    for b0 in range(0, nf, batch):  # stage 3: every score of the row in float64, exact selection
        ids = rows[b0:b0 + batch].contiguous()
        h = score_rows(sources, n, offsets, groups, group_weights, norms, ids)
        idx, _, count = dense_rows_topk(h, k, min_similarity, exclude_self=False)
        cand3 = TopK(idx, torch.zeros(idx.shape, dtype=torch.float32, device=dev), count)
        rescore_topk(sources, n, offsets, groups, group_weights, norms, cand3, k, min_similarity, out, q_row0=q_row0,
                     row_ids=ids, margin=float("-inf"))
        del h
    return out


def gram(z_all: torch.Tensor, *, col0: int = 0, width: int | None = None, q_row0: int = 0,
         nq: int | None = None, n: int | None = None, z_q: torch.Tensor | None = None) -> torch.Tensor:
    """Dense (nq, n) fp32 matrix z[q rows, cols] @ z[:, cols].T over the column
    range [col0, col0 + width) of z (default: all columns); ``z_q`` gives the query rows as a
    separate (nq, ld) tensor.  For small nq x n only."""
    _require_cuda(z_all, "z_all")
    if z_all.dim() != 2 or not z_all.is_contiguous():
        raise ValueError("z_all must be a contiguous 2-D tensor")
    n = z_all.shape[0] if n is None else int(n)
    nq = n - q_row0 if nq is None else int(nq)
    ld = z_all.shape[1]
    width = ld - col0 if width is None else int(width)
    if col0 % 4 or width % 4 or col0 < 0 or col0 + width > ld:
        raise ValueError("column ranges must be multiples of 4 floats inside the row")
    if z_q is not None:
        _require_cuda(z_q, "z_q")
        if z_q.dim() != 2 or not z_q.is_contiguous() or z_q.shape[1] != ld:
            raise ValueError("z_q must be contiguous with the same row length as z_all")
        nq, q_row0 = z_q.shape[0], 0
    if q_row0 < 0 or (z_q is None and q_row0 + nq > z_all.shape[0]) or not 1 <= n <= z_all.shape[0]:
        raise ValueError("rows outside z_all")
    out = torch.empty((nq, n), dtype=torch.float32, device=z_all.device)
    lib = native.load()
    base = z_all.data_ptr() + col0 * 4
    q_base = base + q_row0 * ld * 4 if z_q is None else z_q.data_ptr() + col0 * 4
    with torch.cuda.device(z_all.device):
        rc = lib.tvbf_gram_f32(q_base, nq, base, n, width, ld, out.data_ptr(), n, _stream_ptr())
    native.check(rc, "tvbf_gram_f32")
    return out


def weighted_sum3(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, wa: float, wb: float, wc: float):
    """(wa*a + wb*b) + wc*c in float64 without fused multiply-adds."""
    for t, name in ((a, "a"), (b, "b"), (c, "c")):
        _require_cuda(t, name, torch.float64)
        if not t.is_contiguous() or t.shape != a.shape:
            raise ValueError("inputs must be contiguous and of one shape")
    out = torch.empty_like(a)
    lib = native.load()
    with torch.cuda.device(a.device):
        rc = lib.tvbf_weighted_sum3_f64(a.data_ptr(), b.data_ptr(), c.data_ptr(), float(wa), float(wb), float(wc),
                                        out.data_ptr(), a.numel(), _stream_ptr())
    native.check(rc, "tvbf_weighted_sum3_f64")
    return out


def upper_tri_stats(m: torch.Tensor) -> torch.Tensor:
    """{mean, std, min, max, median} of the entries j > i of a square float64 matrix (5 doubles on the GPU)."""
    _require_cuda(m, "matrix", torch.float64)
    if m.dim() != 2 or m.shape[0] != m.shape[1] or m.stride(1) != 1:
        raise ValueError("matrix must be square with contiguous rows")
    lib = native.load()
    out = torch.empty(5, dtype=torch.float64, device=m.device)
    ws_bytes = int(lib.tvbf_stats_workspace_bytes())
    ws = torch.empty(ws_bytes // 8, dtype=torch.int64, device=m.device)
    with torch.cuda.device(m.device):
        rc = lib.tvbf_upper_tri_stats_f64(m.data_ptr(), m.shape[0], m.stride(0), out.data_ptr(), ws.data_ptr(),
                                          ws_bytes, _stream_ptr())
    native.check(rc, "tvbf_upper_tri_stats_f64")
    return out


def gram_stats(z: torch.Tensor, n: int | None = None, info: dict | None = None) -> torch.Tensor:
    """{mean, std, min, max, median} of the strict upper triangle of ``z[:n] @ z[:n].T`` (5 doubles
    on the GPU); the matrix is never stored.  ``z`` is (rows >= n, ld) fp32 with ld % 32 == 0.
    ``info`` (tests, tuning): filled with what the call did -- synchronises."""
    _require_cuda(z, "z")
    if z.dim() != 2 or not z.is_contiguous():
        raise ValueError("z must be a contiguous 2-D tensor")
    n = z.shape[0] if n is None else int(n)
    if not 2 <= n <= z.shape[0]:
        raise ValueError("statistics of the upper triangle need 2 <= n <= rows of z")
    lib = native.load()
    out = torch.empty(5, dtype=torch.float64, device=z.device)
    ws_bytes = int(lib.tvbf_gram_stats_workspace_bytes(n))
    ws = torch.empty(ws_bytes // 8, dtype=torch.int64, device=z.device)
    with torch.cuda.device(z.device):
        rc = lib.tvbf_gram_stats_f32(z.data_ptr(), n, z.shape[1], z.shape[1], out.data_ptr(), ws.data_ptr(),
                                     ws_bytes, _stream_ptr())
    native.check(rc, "tvbf_gram_stats_f32")
    if info is not None:
        import ctypes

        six = (ctypes.c_int64 * 6)()
        with torch.cuda.device(z.device):
            native.check(lib.tvbf_gram_stats_inspect(ws.data_ptr(), six, _stream_ptr()), "tvbf_gram_stats_inspect")
        info.update(radix_sweeps=int(six[0]), below=int(six[1]), equal_lo=int(six[2]), buffered=int(six[3]),
                    equal_hi=int(six[4]), capacity=int(six[5]), workspace_bytes=ws_bytes)
    return out


def dense_rows_topk(h: torch.Tensor, k: int, min_similarity: float, *, row0: int = 0, rows: int | None = None,
                    exclude_self: bool = True):
    """Top-k of rows ``row0 .. row0+rows`` of a dense float64 score matrix (n, n) on the GPU.
    Returns (idx int32 (rows,k), score float64 (rows,k), count int32 (rows,))."""
    _require_cuda(h, "matrix", torch.float64)
    if h.dim() != 2 or h.stride(1) != 1:
        raise ValueError("matrix must be 2-D with contiguous rows")
    n_all, n_cols = h.shape
    rows = n_all - row0 if rows is None else int(rows)
    if row0 < 0 or rows < 0 or row0 + rows > n_all:
        raise ValueError("rows outside the matrix")
    if k < 1:
        raise ValueError("k must be >= 1")
    dev = h.device
    idx = torch.empty((rows, k), dtype=torch.int32, device=dev)
    score = torch.empty((rows, k), dtype=torch.float64, device=dev)
    count = torch.empty((rows,), dtype=torch.int32, device=dev)
    lib = native.load()
    with torch.cuda.device(dev):
        rc = lib.tvbf_dense_rows_topk_f64(h.data_ptr() + row0 * h.stride(0) * 8, rows, n_cols, h.stride(0), row0, k,
                                          float(min_similarity), 1 if exclude_self else 0, idx.data_ptr(),
                                          score.data_ptr(), count.data_ptr(), _stream_ptr())
    native.check(rc, "tvbf_dense_rows_topk_f64")
    return idx, score, count


def query_topk(z_q: torch.Tensor, z_all: torch.Tensor, k: int, min_similarity: float, *,
               self_index=None, n: int | None = None, workspace: torch.Tensor | None = None) -> TopK:
    """Neighbours of a few query vectors (1..64 rows of ``z_q``) among the first ``n`` rows of
    ``z_all``: one memory-bound pass over ``z_all``.  ``self_index[i]`` is the column left out
    for query i (-1 or None: nothing)."""
    _require_cuda(z_all, "z_all")
    _require_cuda(z_q, "z_q")
    if z_all.dim() != 2 or not z_all.is_contiguous() or z_q.dim() != 2 or z_q.stride(1) != 1:
        raise ValueError("z_all must be contiguous 2-D and z_q 2-D with contiguous rows")
    if z_q.shape[1] != z_all.shape[1]:
        raise ValueError("z_q and z_all must have the same row length")
    nq = z_q.shape[0]
    n = z_all.shape[0] if n is None else int(n)
    if not 1 <= n <= z_all.shape[0]:
        raise ValueError("n outside z_all")
    if not 1 <= nq <= native.TVBF_MAX_QUERIES:
        raise ValueError(f"1..{native.TVBF_MAX_QUERIES} queries per call")
    if not 1 <= k <= native.TVBF_MAX_K:
        raise ValueError(f"k must be in 1..{native.TVBF_MAX_K}")
    selfs = None
    if self_index is not None:
        vals = [int(v) for v in self_index]
        if len(vals) != nq:
            raise ValueError("self_index needs one entry per query")
        selfs = (ctypes.c_int64 * nq)(*vals)
    dev = z_all.device
    out = TopK(torch.empty((nq, k), dtype=torch.int32, device=dev),
               torch.empty((nq, k), dtype=torch.float32, device=dev),
               torch.empty((nq,), dtype=torch.int32, device=dev))
    lib = native.load()
    with torch.cuda.device(dev):
        need = int(lib.tvbf_query_workspace_bytes(nq, n, k))
        if workspace is None:
            workspace = torch.empty((need + 15) // 16 * 2, dtype=torch.int64, device=dev)
        ws_bytes = workspace.numel() * workspace.element_size()
        rc = lib.tvbf_query_topk_f32(z_q.data_ptr(), nq, z_q.stride(0) if nq > 1 else z_q.shape[1], selfs,
                                     z_all.data_ptr(), n, z_all.shape[1], z_all.shape[1], k, float(min_similarity),
                                     out.idx.data_ptr(), out.score.data_ptr(), out.count.data_ptr(),
                                     workspace.data_ptr(), ws_bytes, _stream_ptr())
    native.check(rc, "tvbf_query_topk_f32")
    return out


class TopkSession:
    """Top-k of a block of query rows built up from column pieces.

    begin -> accumulate(piece) ... -> finish.  A piece is one or two row ranges of
    normalised features given as ``(tensor, n_rows, first_global_column)``; their
    rows are the columns of the score matrix.  Every accumulate is one fused launch
    on torch's current stream; lists of all pieces are merged by ``finish``.  The
    result equals a single :func:`cosine_topk` over all columns, bit for bit.
    """

    def __init__(self, z_q: torch.Tensor, nq: int, q_row0: int, k: int, min_similarity: float, *,
                 exclude_self: bool = True, slots: int = 32, workspace: torch.Tensor | None = None):
        _require_cuda(z_q, "z_q")
        if z_q.dim() != 2 or not z_q.is_contiguous() or nq > z_q.shape[0]:
            raise ValueError("z_q must be a contiguous 2-D tensor with at least nq rows")
        if not 1 <= k <= native.TVBF_MAX_K:
            raise ValueError(f"k must be in 1..{native.TVBF_MAX_K}")
        self.lib = native.load()
        self.z_q, self.nq, self.q_row0, self.k = z_q, int(nq), int(q_row0), int(k)
        self.min_similarity, self.exclude_self = float(min_similarity), bool(exclude_self)
        self.ld = z_q.shape[1]
        self.dev = z_q.device
        self.slots = int(slots)  # what the workspace holds
        self.max_lists = int(self.lib.tvbf_topk_max_slots(k))  # what one merge takes
        self.gated_used = 0
        need = int(self.lib.tvbf_topk_workspace_bytes_for_slots(max(self.nq, 1), k, self.slots))
        if workspace is None:
            workspace = torch.empty((need + 15) // 16 * 2, dtype=torch.int64, device=self.dev)
        self.workspace = workspace
        self.ws_bytes = workspace.numel() * workspace.element_size()
        if self.ws_bytes < need:
            raise ValueError(f"workspace holds {self.ws_bytes} bytes, {need} needed for {self.slots} slots")
        self.used = 0
        self.tiles = 0  # 128 x 128 score tiles the launches so far compute (gated re-runs not counted)
        with torch.cuda.device(self.dev):
            rc = self.lib.tvbf_topk_begin(self.nq, k, self.min_similarity, workspace.data_ptr(), self.ws_bytes,
                                          _stream_ptr())
        native.check(rc, "tvbf_topk_begin")

    @staticmethod
    def _range(r, ld):
        if r is None:
            return 0, 0, 0, None
        t, n, col0 = r
        n = int(n)
        if n == 0:
            return 0, 0, 0, None
        _require_cuda(t, "column range")
        if t.dim() != 2 or not t.is_contiguous() or t.shape[1] != ld or n > t.shape[0]:
            raise ValueError("a column range must be a contiguous (rows >= n, ld) tensor")
        return t.data_ptr(), n, int(col0), t

    def bounds(self) -> torch.Tensor:
        """The per-row lower bounds of the K-th keys (64-bit keys at the start of the workspace):
        what a shard publishes for the column side of its peers."""
        return self.workspace[: self.nq]

    def accumulate_own_symmetric(self, side: "ColumnSide", max_slots: int = 16) -> int:
        """The block of these rows against their own columns (z_q is both), every pair once; the
        column side goes through ``side`` at the rows' global indices.  Raises when the shape is
        not taken (row length not a multiple of 64, fewer than 16 blocks of 128 rows)."""
        import ctypes

        if self.nq == 0:
            return 0
        budget = min(max_slots, self.slots - self.used, self.max_lists - self.used)
        used = ctypes.c_int(0)
        st = native.ColSide(None, side.state.data_ptr(), side.bytes, side.n_total, side.cap)
        with torch.cuda.device(self.dev):
            rc = self.lib.tvbf_topk_accumulate_sym_f32(
                self.z_q.data_ptr(), self.nq, self.q_row0, self.ld, self.ld, self.k, self.min_similarity,
                1 if self.exclude_self else 0, self.workspace.data_ptr(), self.ws_bytes, self.used, budget,
                ctypes.byref(used), ctypes.byref(st), _stream_ptr())
            self.tiles += int(self.lib.tvbf_topk_sym_tiles(self.nq))
        native.check(rc, "tvbf_topk_accumulate_sym_f32")
        self.used += used.value
        return used.value

    def accumulate_own_matrix(self, matrix: torch.Tensor) -> int:
        """The block of these rows against their own columns through the stored-matrix route (every
        pair once, scores stored as an nq x nq matrix in ``matrix``, one scan per row): one list slot.
        ``matrix``: int64 or float32 device tensor of at least :func:`own_matrix_bytes` bytes."""
        if self.nq == 0:
            return 0
        _require_cuda(matrix, "matrix", None)
        if self.slots - self.used < 1 or self.max_lists - self.used < 1:
            raise ValueError("no partial-list slot left")
        used = ctypes.c_int(0)
        with torch.cuda.device(self.dev):
            rc = self.lib.tvbf_topk_accumulate_matrix_f32(
                self.z_q.data_ptr(), self.nq, self.q_row0, self.ld, self.ld, self.k, self.min_similarity,
                1 if self.exclude_self else 0, self.workspace.data_ptr(), self.ws_bytes, self.used, ctypes.byref(used),
                matrix.data_ptr(), matrix.numel() * matrix.element_size(), _stream_ptr())
        native.check(rc, "tvbf_topk_accumulate_matrix_f32")
        self.used += used.value
        t = (self.nq + 127) // 128
        self.tiles += t * (t + 1) // 2
        return used.value

    def slot_lists(self) -> torch.Tensor:
        """The partial lists written so far: (slots used, nq, k) int64 keys (a view of the workspace)."""
        first = ((self.nq * 8 + 15) // 16 * 16) // 8
        return self.workspace[first : first + self.used * self.nq * self.k].view(self.used, self.nq, self.k)

    def accumulate(self, range_a, range_b=None, max_slots: int = 16, *, colside=None, gate=None) -> int:
        """One fused launch over up to two column ranges.  ``colside=(ColumnSide, col_bound)`` also
        hands the scores to the shows of the columns; ``gate`` (the overflow words of a ColumnSide,
        int32 device tensor) restricts the sweep to the 128-row blocks that hold a flagged show --
        the slots are reserved either way."""
        pa, na, ca, keep_a = self._range(range_a, self.ld)
        pb, nb, cb, keep_b = self._range(range_b, self.ld)
        if na + nb == 0 or self.nq == 0:
            return 0
        # the lists of the ordinary sweeps are merged together, those of the gated sweeps instead of them
        merged_so_far = self.gated_used if gate is not None else self.used
        budget = min(max_slots, self.slots - self.used, self.max_lists - merged_so_far)
        if budget < 1:
            raise ValueError("no partial-list slots left: create the session with more slots, or K is too large "
                             "for this many sweeps")
        import ctypes

        used = ctypes.c_int(0)
        cs_ref = None
        if colside is not None:
            side, col_bound = colside
            cs_ref = ctypes.byref(side.struct(col_bound))
        gate_ptr = None
        if gate is not None:
            _require_cuda(gate, "gate", dtype=torch.int32)
            gate_ptr = gate.data_ptr()
        with torch.cuda.device(self.dev):
            rc = self.lib.tvbf_topk_accumulate_ex_f32(
                self.z_q.data_ptr(), self.nq, self.q_row0, self.ld, self.ld, pa, na, ca, pb, nb, cb, self.k,
                self.min_similarity, 1 if self.exclude_self else 0, self.workspace.data_ptr(), self.ws_bytes,
                self.used, budget, ctypes.byref(used), cs_ref, gate_ptr, _stream_ptr())
        native.check(rc, "tvbf_topk_accumulate_ex_f32")
        self.used += used.value
        if gate is not None:
            self.gated_used += used.value
        else:
            self.tiles += ((self.nq + 127) // 128) * ((na + 127) // 128 + (nb + 127) // 128)
        return used.value

    def finish(self, out: TopK | None = None, *, extra: torch.Tensor | None = None, gate=None,
               alt: tuple[int, int] | None = None, slots: int | None = None) -> TopK:
        """Merge the partial lists of slots 0 .. slots-1 (default: all used) and of ``extra``
        (int64 keys, (lists, rows >= nq, k), every list contiguous, any distance between lists).  With ``gate`` (overflow words as for
        accumulate) and ``alt=(first, count)`` a flagged row merges slots first .. first+count-1 alone."""
        if out is None:
            out = TopK(torch.empty((self.nq, self.k), dtype=torch.int32, device=self.dev),
                       torch.empty((self.nq, self.k), dtype=torch.float32, device=self.dev),
                       torch.empty((self.nq,), dtype=torch.int32, device=self.dev))
        if self.nq == 0:  # a rank without rows: nothing to merge
            return out
        slots = self.used if slots is None else int(slots)
        n_extra, stride, extra_ptr = 0, 0, None
        if extra is not None and extra.numel():
            _require_cuda(extra, "extra", dtype=torch.int64)
            if (extra.dim() != 3 or extra.shape[2] != self.k or extra.shape[1] < self.nq or extra.stride(2) != 1
                    or extra.stride(1) != self.k or (extra.shape[0] > 1 and extra.stride(0) < extra.shape[1] * self.k)):
                raise ValueError("extra must be (lists, rows >= nq, k) int64 with contiguous lists")
            n_extra, stride, extra_ptr = extra.shape[0], extra.stride(0), extra.data_ptr()
        gate_ptr, alt_first, alt_slots = None, 0, 0
        if gate is not None:
            _require_cuda(gate, "gate", dtype=torch.int32)
            gate_ptr, (alt_first, alt_slots) = gate.data_ptr() + 4 * (4 + self.q_row0), alt
        with torch.cuda.device(self.dev):
            rc = self.lib.tvbf_topk_finish_ex(self.nq, self.k, slots, extra_ptr, n_extra, stride, gate_ptr,
                                              alt_first, alt_slots, out.idx.data_ptr(), out.score.data_ptr(),
                                              out.count.data_ptr(), self.workspace.data_ptr(), self.ws_bytes,
                                              _stream_ptr())
        native.check(rc, "tvbf_topk_finish_ex")
        return out


class ColumnSide:
    """Device state of the column side of a sharded symmetric sweep: per show of the whole
    catalogue a counter and a buffer of ``cap`` candidates (score, global row of the query side),
    behind one overflow flag."""

    def __init__(self, n_total: int, device, cap: int = 1024):
        self.lib = native.load()
        self.n_total, self.cap, self.dev = int(n_total), int(cap), torch.device(device)
        self.bytes = int(self.lib.tvbf_topk_colside_bytes(self.n_total, self.cap))
        if not self.bytes:
            raise ValueError("n_total >= 1 and 1 <= cap <= 2048")
        self.state = torch.empty((self.bytes + 15) // 16 * 2, dtype=torch.int64, device=self.dev)
        self._struct = None
        self._clean = False
        self.begin()

    def begin(self) -> None:
        """Ready for a new job: the first time everything is cleared, afterwards only what the
        previous job wrote (its counters say how much)."""
        fn = self.lib.tvbf_topk_colside_reset if self._clean else self.lib.tvbf_topk_colside_begin
        with torch.cuda.device(self.dev):
            rc = fn(self.state.data_ptr(), self.bytes, self.n_total, self.cap, _stream_ptr())
        native.check(rc, "tvbf_topk_colside_begin / reset")
        self._clean = True

    @property
    def flags(self) -> torch.Tensor:
        """The overflow words (int32 view): [0] some buffer overflowed, [4 + show] this show's did."""
        return self.state.view(torch.int32)[: 4 + self.n_total]

    def struct(self, col_bound: torch.Tensor):
        _require_cuda(col_bound, "col_bound", dtype=torch.int64)
        if col_bound.dim() != 1 or col_bound.numel() < self.n_total or not col_bound.is_contiguous():
            raise ValueError("col_bound must hold n_total contiguous keys")
        self._struct = native.ColSide(col_bound.data_ptr(), self.state.data_ptr(), self.bytes, self.n_total, self.cap)
        self._keep = col_bound
        return self._struct

    def reduce(self, col_bound: torch.Tensor, row0: int, rows: int, k: int, min_similarity: float,
               out: torch.Tensor) -> torch.Tensor:
        """The K best buffered candidates of shows row0 .. row0+rows-1 as partial lists into
        ``out`` ((>= rows, k) int64 keys, contiguous)."""
        import ctypes

        _require_cuda(out, "out", dtype=torch.int64)
        if out.dim() != 2 or out.shape[1] != k or out.shape[0] < rows or not out.is_contiguous():
            raise ValueError("out must be a contiguous (>= rows, k) int64 tensor")
        with torch.cuda.device(self.dev):
            rc = self.lib.tvbf_topk_colside_reduce(ctypes.byref(self.struct(col_bound)), int(row0), int(rows), int(k),
                                                   float(min_similarity), out.data_ptr(), _stream_ptr())
        native.check(rc, "tvbf_topk_colside_reduce")
        return out
