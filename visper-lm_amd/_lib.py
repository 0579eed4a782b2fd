"""ctypes binding of libvisper_hip.so (the C ABI declared in include/visper_hip.h).

The product path has NO CPU fallback: if the library is missing, or a call returns non-zero,
this module raises.  `import torch` happens first so the HIP runtime that PyTorch already loaded
(libamdhip64.so.7) is the one the kernels launch on (same streams, same device memory)."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the library: shares its HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VP_LIB_PATH") or os.path.join(_HERE, "libvisper_hip.so")     # VP_LIB_PATH: dev aid (A/B of build variants)
# the -DVP_DEBUG build of the same sources (measurement entry points vp_debug_*, include/visper_hip_debug.h): never on the product path;
# bench.py's shader-clock probe, two co-residency tests and tools/ load it explicitly (debug_library())
DEBUG_LIB_PATH = os.path.join(_HERE, "libvisper_hip_debug.so")

i, l, f, p = C.c_int, C.c_long, C.c_float, C.c_void_p

# name -> argtypes (every function returns int unless listed in _RET)
_SIGS = {
    "vp_version": [],
    "vp_device_info": [p, p, p],
    "vp_gemm_bf16": [i, i, i, p, l, p, l, p, l, p, p, l, i, i, i, p, p],
    "vp_gemm_sched_workspace_bytes": [],
    "vp_gemm_bf16_swiglu": [i, i, i, i, p, l, p, l, p, l, p, l, p, l, p, p],
    "vp_gemm_bf16_rope": [i, i, i, p, l, p, l, p, l, p, i, p, p, p, i, p],
    "vp_gemm_bf16_sumsq": [i, i, i, p, l, p, l, p, l, p, l, p, p],
    "vp_rstd_from_sumsq": [i, i, p, i, f, p, p],
    "vp_gemm_tn_bf16": [i, i, i, p, l, p, l, p, l, i, i, p, p],
    "vp_transpose_bf16": [i, i, p, l, p, l, p],
    "vp_transpose_batched_bf16": [i, i, i, p, l, l, p, l, l, p],
    "vp_rmsnorm_fwd": [i, i, p, l, p, f, p, l, p, p],
    "vp_rmsnorm_bwd": [i, i, p, p, p, p, p, p, l, p],
    "vp_layernorm_fwd": [i, i, p, l, p, p, f, p, l, p, p, p],
    "vp_layernorm_bwd_dx": [i, i, p, p, p, p, p, p, p, l, p],
    "vp_layernorm_bwd_wb_partial": [i, i, p, p, p, p, p, p, l, i, p],
    "vp_dwconv7x7_nhwc": [i, i, i, i, p, p, p, p, p],
    "vp_rope": [l, i, i, i, p, l, p, p, p, i, p],
    "vp_im2col3x3_nhwc": [i, i, i, i, i, i, p, p, p],
    "vp_bilinear_nhwc": [i, i, i, i, i, i, i, p, p, p],
    "vp_pixel_shuffle_nhwc": [i, i, i, i, i, p, p, p],
    "vp_minmax_norm": [i, l, p, p, p],
    "vp_swiglu_fwd": [l, i, p, l, p, l, i, p],
    "vp_scatter_add_rows": [l, i, p, l, p, p, p],
    "vp_swiglu_bwd": [l, i, p, l, p, p, l, i, p],
    "vp_act_fwd": [i, l, p, p, p],
    "vp_act_bwd": [i, l, p, p, p, p],
    "vp_add_bf16": [l, p, p, p, p],
    "vp_add2d_bf16": [l, i, p, l, p, l, p],
    "vp_copy2d_bf16": [l, i, p, l, p, l, p],
    "vp_memset_zero": [p, l, p],
    "vp_colsum_partial": [l, i, p, l, p, i, p],
    "vp_colsum_finish": [i, i, p, p, f, i, p],
    "vp_gather_rows": [l, i, p, p, i, p, p, p, l, p],
    "vp_gather_sum_rows": [l, i, i, p, l, i, p, f, p, l, i, i, p],
    "vp_cast_f32_to_bf16": [l, p, p, p],
    "vp_cast_bf16_to_f32": [l, p, p, i, p],
    "vp_scatter_rows_bf16_to_f32": [l, i, p, l, p, p, l, p],
    "vp_sum_f32": [l, p, p, f, p],
    "vp_sumsq_f32": [l, p, p, p, p],
    "vp_attn_fwd": [i, i, i, i, i, i, p, l, l, p, l, l, p, l, l, p, l, l, p, p, i, i, f, p],
    "vp_attn_fwd_bias": [i, i, i, i, i, i, p, l, l, p, l, l, p, l, l, p, l, l, p, p, i, i, f, p, p, i, p],
    "vp_attn_bwd": [i, i, i, i, i, i, p, l, l, p, l, l, p, l, l, p, l, l, p, p, l, l, p, l, l, p, l, l, p, l, l, p, p,
                    i, i, f, p],
    "vp_attn_bwd_rope": [i, i, i, i, i, i, p, l, l, p, l, l, p, l, l, p, l, l, p, p, l, l, p, l, l, p, l, l, p, l, l, p, p,
                    i, i, f, p, p, p, p],
    "vp_ce_fwd_bwd": [l, i, p, l, p, p, f, i, p],
    "vp_emb_loss_workspace": [i, i, l],
    "vp_sumsq_nblk": [l],
    "vp_emb_loss_counter_bytes": [],
    "vp_emb_loss_fwd": [i, i, l, i, p, p, p, p, f, p, p, p, p, p],
    "vp_emb_loss_bwd": [i, i, l, i, p, p, p, f, p, p],
    "vp_emb_loss_fwd_multi": [i, i, i, p, i, p, p, p, p, p, p, p, p, p, p],
    "vp_emb_loss_bwd_multi": [i, i, i, p, i, p, p, p, p, p, p],
    "vp_adamw": [l, p, p, p, p, p, f, f, f, f, f, i, f, p],
    "vp_comm_unique_id_bytes": [],
    "vp_comm_unique_id": [p],
    "vp_comm_init": [i, i, p, p],
    "vp_comm_allreduce_async": [p, p, l, i, p],
    "vp_comm_wait": [p, p],
    "vp_comm_allgather": [p, p, p, l, i, p],
    "vp_comm_info": [p, p, p],
    "vp_comm_destroy": [p],
}
_RET_LONG = {"vp_emb_loss_workspace", "vp_gemm_sched_workspace_bytes", "vp_emb_loss_counter_bytes"}
EXPORTS = ["vp_last_error_string"] + list(_SIGS)
# measurement / development entry points (include/visper_hip_debug.h): present only in a -DVP_DEBUG build; bound when the library has them
_DEBUG_SIGS = {
    "vp_debug_occupy": [i, l, p],
    "vp_debug_gemm_flags": [i],
    "vp_debug_stamps": [p],
    "vp_debug_attn_stamps": [p],
    "vp_debug_emb_loss_stamps": [p],
}

_lib = None
_dbg = None


def _bind(path):
    lib = C.CDLL(path)
    lib.vp_last_error_string.restype = C.c_char_p
    lib.vp_last_error_string.argtypes = []
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_long if name in _RET_LONG else C.c_int
    for name, args in _DEBUG_SIGS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, C.c_int
    return lib


def load():
    """Load the library (raises if it has not been built: run `python __graft_entry__.py` / build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found — the HIP extension is not built (no CPU fallback exists). "
                               "Run __graft_entry__.build().")
        _lib = _bind(LIB_PATH)
    return _lib


import contextlib  # noqa: E402


@contextlib.contextmanager
def debug_library():
    """Inside the block every call of this module goes to the -DVP_DEBUG build (libvisper_hip_debug.so: the same kernels plus the vp_debug_*
    measurement entry points); the product library is restored on exit.  If the product path itself is a debug build (VP_LIB_PATH) it is used."""
    global _lib, _dbg
    prev = load()
    if hasattr(prev, "vp_debug_gemm_flags"):
        yield prev
        return
    if _dbg is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise RuntimeError(f"{DEBUG_LIB_PATH} not found: build it with `make -C visper-lm_amd/csrc debug`")
        _dbg = _bind(DEBUG_LIB_PATH)
    _lib = _dbg
    try:
        yield _dbg
    finally:
        _lib = prev


def call(name, *args):
    lib = load()
    if name in _DEBUG_SIGS and not hasattr(lib, name):
        raise RuntimeError(f"{name}: this libvisper_hip.so was built without -DVP_DEBUG (make VP_DEBUG=1)")
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.vp_last_error_string().decode()}")
    return rc


def raw(name, *args):
    """Call returning the raw int (for query functions such as vp_emb_loss_workspace / vp_version)."""
    return getattr(load(), name)(*args)
