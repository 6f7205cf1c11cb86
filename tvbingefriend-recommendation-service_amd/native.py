"""ctypes binding of the C ABI declared in include/tvbf.h.

The shared library is built in-tree (csrc/libtvbf.so) by :func:`build`.  There
is no fallback: if the library is missing or a call fails, an exception is
raised.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
INCLUDE = Path(__file__).resolve().parent.parent / "include"
# TVBF_LIB selects another build of the same ABI (timing-only diagnostic variants under build/)
LIB_PATH = Path(os.environ["TVBF_LIB"]) if os.environ.get("TVBF_LIB") else CSRC / "libtvbf.so"
SOURCES = [CSRC / "tvbf_kernels.hip", CSRC / "tvbf_stats.hip", CSRC / "tvbf_assemble.hip", CSRC / "tvbf_rescore.hip"]

# Every symbol include/tvbf.h declares.
SYMBOLS = (
    "tvbf_version",
    "tvbf_last_error",
    "tvbf_normalize_blocks_f32",
    "tvbf_assemble_normalize_f32",
    "tvbf_topk_workspace_bytes",
    "tvbf_cosine_topk_f32",
    "tvbf_cosine_topk_static_f32",
    "tvbf_cosine_topk_sym_workspace_bytes",
    "tvbf_cosine_topk_sym_f32",
    "tvbf_cosine_topk_sym_inspect",
    "tvbf_topk_sym_tiles",
    "tvbf_cosine_topk_sym_tiles",
    "tvbf_topk_max_slots",
    "tvbf_topk_workspace_bytes_for_slots",
    "tvbf_topk_begin",
    "tvbf_topk_accumulate_f32",
    "tvbf_topk_finish",
    "tvbf_topk_matrix_bytes",
    "tvbf_topk_accumulate_matrix_f32",
    "tvbf_topk_colside_bytes",
    "tvbf_topk_colside_begin",
    "tvbf_topk_colside_reset",
    "tvbf_topk_accumulate_ex_f32",
    "tvbf_topk_accumulate_sym_f32",
    "tvbf_topk_colside_reduce",
    "tvbf_topk_finish_ex",
    "tvbf_query_workspace_bytes",
    "tvbf_query_topk_f32",
    "tvbf_gram_f32",
    "tvbf_gram_stats_workspace_bytes",
    "tvbf_gram_stats_f32",
    "tvbf_gram_stats_inspect",
    "tvbf_component_scores_f32",
    "tvbf_refine_min_similarity_f64",
    "tvbf_block_norms_f64",
    "tvbf_rescore_topk_f64",
    "tvbf_score_rows_f64",
    "tvbf_band_candidates_f32",
    "tvbf_weighted_sum3_f64",
    "tvbf_stats_workspace_bytes",
    "tvbf_upper_tri_stats_f64",
    "tvbf_dense_rows_topk_f64",
)

TVBF_MAX_BLOCKS = 8
TVBF_MAX_K = 116
TVBF_MAX_SOURCES = 8
TVBF_MAX_QUERIES = 64
TVBF_STATIC_SCRATCH_BYTES = 32 << 20
SRC_DENSE_F32, SRC_DENSE_F64, SRC_DENSE_I64, SRC_DENSE_U8, SRC_CSR_F32, SRC_CSR_F64 = range(6)


class Source(ctypes.Structure):
    """struct tvbf_source of include/tvbf.h."""

    _fields_ = [("kind", ctypes.c_int32), ("index64", ctypes.c_int32), ("width", ctypes.c_int32),
                ("z_col", ctypes.c_int32), ("data", ctypes.c_void_p), ("ld", ctypes.c_int64),
                ("indptr", ctypes.c_void_p), ("indices", ctypes.c_void_p)]


class ColSide(ctypes.Structure):
    """struct tvbf_colside of include/tvbf.h."""

    _fields_ = [("col_bound", ctypes.c_void_p), ("state", ctypes.c_void_p), ("state_bytes", ctypes.c_size_t),
                ("n_total", ctypes.c_int64), ("cap", ctypes.c_int32)]


class NativeLibraryError(RuntimeError):
    """The HIP library is missing, cannot be loaded, or a call into it failed."""


def hipcc_command(out: Path) -> list[str]:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        # a*b+c is fused only where the source says fma(): float64 results are compared bitwise
        "-ffp-contract=off",
        f"-I{INCLUDE}",
        # packed fp32 VALU forms are switched off for device code
        "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
        "-o", str(out), *[str(s) for s in SOURCES],
    ]


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into csrc/libtvbf.so (no GPU needed)."""
    deps = SOURCES + [INCLUDE / "tvbf.h"] + sorted((CSRC / "kernels").glob("*.inc"))  # tvbf_kernels.hip includes them
    if not force and LIB_PATH.exists() and all(LIB_PATH.stat().st_mtime >= p.stat().st_mtime for p in deps):
        return LIB_PATH
    tmp = LIB_PATH.with_suffix(".so.tmp")
    cmd = hipcc_command(tmp)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise NativeLibraryError(f"hipcc failed ({res.returncode}):\n{' '.join(cmd)}\n{res.stderr}")
    if verbose:
        print(res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


RECORDS_EXT_PATH = CSRC / "_tvbf_records.so"  # CPython extension: record dicts of a top-K result (host formatting)


def build_records_ext(force: bool = False):
    """Compile csrc/tvbf_records.c with the host compiler into csrc/_tvbf_records.so.  Returns the
    path, or None when the interpreter's headers or a C compiler are missing (the Python loop of
    scripts.populate_database.neighbour_records then does the same work)."""
    import shutil
    import sysconfig

    src = CSRC / "tvbf_records.c"
    inc = sysconfig.get_paths().get("include")
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc or not inc or not (Path(inc) / "Python.h").exists():
        return None
    if not force and RECORDS_EXT_PATH.exists() and RECORDS_EXT_PATH.stat().st_mtime >= src.stat().st_mtime:
        return RECORDS_EXT_PATH
    tmp = RECORDS_EXT_PATH.with_suffix(".so.tmp")
    res = subprocess.run([cc, "-O2", "-fPIC", "-shared", "-Wall", f"-I{inc}", "-o", str(tmp), str(src)],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise NativeLibraryError(f"{cc} failed on {src}:\n{res.stderr}")
    os.replace(tmp, RECORDS_EXT_PATH)
    return RECORDS_EXT_PATH


_lib = None


def load() -> ctypes.CDLL:
    """Load csrc/libtvbf.so and declare the signatures of include/tvbf.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)"
        )
    try:
        lib = ctypes.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    c = ctypes
    i64, i32, f32, vp, sz = c.c_int64, c.c_int32, c.c_float, c.c_void_p, c.c_size_t
    lib.tvbf_version.restype = c.c_int
    lib.tvbf_version.argtypes = []
    lib.tvbf_last_error.restype = c.c_char_p
    lib.tvbf_last_error.argtypes = []
    lib.tvbf_normalize_blocks_f32.restype = c.c_int
    lib.tvbf_normalize_blocks_f32.argtypes = [vp, i64, i64, i64, c.POINTER(i32), c.POINTER(f32), c.c_int,
                                              vp, i64, vp]
    lib.tvbf_assemble_normalize_f32.restype = c.c_int
    lib.tvbf_assemble_normalize_f32.argtypes = [c.POINTER(Source), c.c_int, i64, i64, c.POINTER(i32), c.POINTER(f32),
                                                c.c_int, vp, i64, vp]
    lib.tvbf_topk_workspace_bytes.restype = sz
    lib.tvbf_topk_workspace_bytes.argtypes = [i64, i64, i64, c.c_int]
    lib.tvbf_cosine_topk_f32.restype = c.c_int
    lib.tvbf_cosine_topk_f32.argtypes = [vp, i64, i64, vp, i64, i64, i64, c.c_int, f32, c.c_int,
                                         vp, vp, vp, vp, sz, vp]
    lib.tvbf_cosine_topk_static_f32.restype = c.c_int
    lib.tvbf_cosine_topk_static_f32.argtypes = [vp, i64, i64, vp, i64, i64, i64, c.c_int, f32, c.c_int, vp, vp, vp, vp]
    lib.tvbf_cosine_topk_sym_workspace_bytes.restype = sz
    lib.tvbf_cosine_topk_sym_workspace_bytes.argtypes = [i64, i64, c.c_int]
    lib.tvbf_cosine_topk_sym_f32.restype = c.c_int
    lib.tvbf_cosine_topk_sym_f32.argtypes = [vp, i64, i64, i64, c.c_int, f32, c.c_int, vp, vp, vp, vp, sz, vp]
    lib.tvbf_cosine_topk_sym_tiles.restype = i64
    lib.tvbf_cosine_topk_sym_tiles.argtypes = [i64, i64, c.c_int]
    lib.tvbf_topk_sym_tiles.restype = i64
    lib.tvbf_topk_sym_tiles.argtypes = [i64]
    lib.tvbf_cosine_topk_sym_inspect.restype = c.c_int
    lib.tvbf_cosine_topk_sym_inspect.argtypes = [vp, i64, i64, c.c_int, c.POINTER(i64), vp]
    lib.tvbf_topk_matrix_bytes.restype = sz
    lib.tvbf_topk_matrix_bytes.argtypes = [i64, i64]
    lib.tvbf_topk_accumulate_matrix_f32.restype = c.c_int
    lib.tvbf_topk_accumulate_matrix_f32.argtypes = [vp, i64, i64, i64, i64, c.c_int, f32, c.c_int, vp, sz, c.c_int,
                                                    c.POINTER(c.c_int), vp, sz, vp]
    lib.tvbf_topk_colside_bytes.restype = sz
    lib.tvbf_topk_colside_bytes.argtypes = [i64, c.c_int]
    lib.tvbf_topk_colside_begin.restype = c.c_int
    lib.tvbf_topk_colside_begin.argtypes = [vp, sz, i64, c.c_int, vp]
    lib.tvbf_topk_colside_reset.restype = c.c_int
    lib.tvbf_topk_colside_reset.argtypes = [vp, sz, i64, c.c_int, vp]
    lib.tvbf_topk_accumulate_ex_f32.restype = c.c_int
    lib.tvbf_topk_accumulate_ex_f32.argtypes = [vp, i64, i64, i64, i64, vp, i64, i64, vp, i64, i64, c.c_int, f32,
                                                c.c_int, vp, sz, c.c_int, c.c_int, c.POINTER(c.c_int),
                                                c.POINTER(ColSide), vp, vp]
    lib.tvbf_topk_accumulate_sym_f32.restype = c.c_int
    lib.tvbf_topk_accumulate_sym_f32.argtypes = [vp, i64, i64, i64, i64, c.c_int, f32, c.c_int, vp, sz, c.c_int,
                                                 c.c_int, c.POINTER(c.c_int), c.POINTER(ColSide), vp]
    lib.tvbf_topk_colside_reduce.restype = c.c_int
    lib.tvbf_topk_colside_reduce.argtypes = [c.POINTER(ColSide), i64, i64, c.c_int, f32, vp, vp]
    lib.tvbf_topk_finish_ex.restype = c.c_int
    lib.tvbf_topk_finish_ex.argtypes = [i64, c.c_int, c.c_int, vp, c.c_int, i64, vp, c.c_int, c.c_int, vp, vp, vp,
                                        vp, sz, vp]
    lib.tvbf_component_scores_f32.restype = c.c_int
    lib.tvbf_component_scores_f32.argtypes = [vp, i64, vp, i64, i64, c.POINTER(i32), c.POINTER(f32), c.c_int,
                                              vp, c.c_int, vp, vp]
    lib.tvbf_refine_min_similarity_f64.restype = c.c_int
    lib.tvbf_refine_min_similarity_f64.argtypes = [vp, vp, vp, vp, i64, c.c_int, c.c_int, c.POINTER(i32),
                                                   c.POINTER(c.c_double), c.c_double, c.c_double, vp]
    f64 = c.c_double
    lib.tvbf_block_norms_f64.restype = c.c_int
    lib.tvbf_block_norms_f64.argtypes = [c.POINTER(Source), c.c_int, i64, i64, c.POINTER(i32), c.c_int, vp, vp, vp]
    lib.tvbf_rescore_topk_f64.restype = c.c_int
    lib.tvbf_rescore_topk_f64.argtypes = [c.POINTER(Source), c.c_int, i64, i64, c.POINTER(i32), c.c_int,
                                          c.POINTER(i32), c.POINTER(f64), vp, vp, i64, i64, vp, vp, vp, c.c_int,
                                          c.c_int, f64, f64, c.c_int, vp, vp, vp, vp, i64, vp, vp]
    lib.tvbf_score_rows_f64.restype = c.c_int
    lib.tvbf_score_rows_f64.argtypes = [c.POINTER(Source), c.c_int, i64, i64, c.POINTER(i32), c.c_int,
                                        c.POINTER(i32), c.POINTER(f64), vp, vp, i64, c.c_int, vp, i64, vp]
    lib.tvbf_band_candidates_f32.restype = c.c_int
    lib.tvbf_band_candidates_f32.argtypes = [vp, i64, vp, i64, i64, i64, vp, c.c_int, vp, vp, vp, vp]
    lib.tvbf_topk_max_slots.restype = c.c_int
    lib.tvbf_topk_max_slots.argtypes = [c.c_int]
    lib.tvbf_topk_workspace_bytes_for_slots.restype = sz
    lib.tvbf_topk_workspace_bytes_for_slots.argtypes = [i64, c.c_int, c.c_int]
    lib.tvbf_topk_begin.restype = c.c_int
    lib.tvbf_topk_begin.argtypes = [i64, c.c_int, f32, vp, sz, vp]
    lib.tvbf_topk_accumulate_f32.restype = c.c_int
    lib.tvbf_topk_accumulate_f32.argtypes = [vp, i64, i64, i64, i64, vp, i64, i64, vp, i64, i64, c.c_int, f32, c.c_int,
                                             vp, sz, c.c_int, c.c_int, c.POINTER(c.c_int), vp]
    lib.tvbf_topk_finish.restype = c.c_int
    lib.tvbf_topk_finish.argtypes = [i64, c.c_int, c.c_int, vp, vp, vp, vp, sz, vp]
    lib.tvbf_query_workspace_bytes.restype = sz
    lib.tvbf_query_workspace_bytes.argtypes = [i64, i64, c.c_int]
    lib.tvbf_query_topk_f32.restype = c.c_int
    lib.tvbf_query_topk_f32.argtypes = [vp, i64, i64, c.POINTER(i64), vp, i64, i64, i64, c.c_int, f32,
                                        vp, vp, vp, vp, sz, vp]
    lib.tvbf_gram_f32.restype = c.c_int
    lib.tvbf_gram_f32.argtypes = [vp, i64, vp, i64, i64, i64, vp, i64, vp]
    lib.tvbf_gram_stats_workspace_bytes.restype = sz
    lib.tvbf_gram_stats_workspace_bytes.argtypes = [i64]
    lib.tvbf_gram_stats_f32.restype = c.c_int
    lib.tvbf_gram_stats_f32.argtypes = [vp, i64, i64, i64, vp, vp, sz, vp]
    lib.tvbf_gram_stats_inspect.restype = c.c_int
    lib.tvbf_gram_stats_inspect.argtypes = [vp, c.POINTER(i64), vp]
    lib.tvbf_weighted_sum3_f64.restype = c.c_int
    lib.tvbf_weighted_sum3_f64.argtypes = [vp, vp, vp, c.c_double, c.c_double, c.c_double, vp, i64, vp]
    lib.tvbf_stats_workspace_bytes.restype = sz
    lib.tvbf_stats_workspace_bytes.argtypes = []
    lib.tvbf_upper_tri_stats_f64.restype = c.c_int
    lib.tvbf_upper_tri_stats_f64.argtypes = [vp, i64, i64, vp, vp, sz, vp]
    lib.tvbf_dense_rows_topk_f64.restype = c.c_int
    lib.tvbf_dense_rows_topk_f64.argtypes = [vp, i64, i64, i64, i64, c.c_int, c.c_double, c.c_int, vp, vp, vp, vp]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tvbf_last_error().decode("utf-8", "replace")
        raise NativeLibraryError(f"{what} failed with code {rc}: {msg}")


def block_arrays(offsets, scales):
    """Host arrays (int32 offsets, float scales) in the form the C ABI reads."""
    nb = len(scales)
    if len(offsets) != nb + 1:
        raise ValueError("offsets must have one more entry than scales")
    if not 1 <= nb <= TVBF_MAX_BLOCKS:
        raise ValueError(f"1..{TVBF_MAX_BLOCKS} feature blocks are supported, got {nb}")
    off = (ctypes.c_int32 * (nb + 1))(*[int(o) for o in offsets])
    sc = (ctypes.c_float * nb)(*[float(s) for s in scales])
    return off, sc, nb
