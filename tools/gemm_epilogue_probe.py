"""Dev tool (gpurun): where a one-wave-per-SIMD GEMM launch spends its time outside the K loop.  In-kernel stamps (vp_debug_gemm_flags 0x10000) of the
LAST of n back-to-back launches: per block, K loop of its first tile, epilogue issue, store drain, whole block; the launch's duration from HIP events.
    python tools/gemm_epilogue_probe.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops, _lib

dev = torch.device("cuda")


def probe(name, M, N, K, fn, n=40, quarter=False, extra=0):
    _lib.call("vp_debug_gemm_flags", extra)
    for _ in range(n - 1):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        fn()
    ev1.record(); torch.cuda.synchronize()
    if extra:
        name += f" [flags {extra:#x}: {ev0.elapsed_time(ev1) * 50:.0f} us per launch back to back]"
    else:
        name += f" [{ev0.elapsed_time(ev1) * 50:.0f} us per launch back to back]"
    _lib.call("vp_debug_gemm_flags", extra | 0x10000 | (0x80000 if quarter else 0))
    try:
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
    finally:
        _lib.call("vp_debug_gemm_flags", 0)
    buf = (C.c_long * 2048)()
    _lib.call("vp_debug_stamps", buf)
    st = np.array(buf[:], dtype=np.int64).reshape(256, 8)
    if quarter:
        st = st[0::4]
        name += " [only a quarter of the CUs working]"
    us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
    tiles = (M // 256) * (N // 256) / 256.0
    loop, epi, drain, whole = us(1, 2), us(2, 3), us(3, 4), us(0, 5)
    t0, t5 = st[:, 0].min(), st[:, 5].max()
    per_xcd_end = [(st[x::8, 5].max() - t0) / 100.0 for x in range(8)] if not quarter else []
    print(f"{name}: {M}x{N}x{K}, {tiles:.2f} tiles per CU; launch {e0.elapsed_time(e1) * 1e3:.0f} us; first..last stamp {(t5 - t0) / 100.0:.0f} us")
    print(f"   per block (mean / max): K loop of tile 0 {loop.mean():.1f} / {loop.max():.1f} us; epilogue issue {epi.mean():.1f} / {epi.max():.1f}; "
          f"store drain {drain.mean():.1f} / {drain.max():.1f}; whole block {whole.mean():.1f} / {whole.max():.1f}")
    print(f"   outside the K loops per tile: {(whole.mean() - tiles * loop.mean()) / tiles:.1f} us = {(1 - tiles * loop.mean() / whole.mean()) * 100:.1f} % of the block; "
          f"XCD end times {[round(x) for x in per_xcd_end]}")


M = 16384
x14 = torch.randn(M, 14336, device=dev, dtype=torch.bfloat16)
x4 = torch.randn(M, 4096, device=dev, dtype=torch.bfloat16)
res = torch.randn(M, 4096, device=dev, dtype=torch.bfloat16)
w_down = torch.randn(4096, 14336, device=dev, dtype=torch.bfloat16) * 0.02
w_o = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16) * 0.02
w_gu = torch.randn(28672, 4096, device=dev, dtype=torch.bfloat16) * 0.02
w_dT = torch.randn(14336, 4096, device=dev, dtype=torch.bfloat16) * 0.02
o4 = torch.empty(M, 4096, device=dev, dtype=torch.bfloat16)
gu = torch.randn(M, 28672, device=dev, dtype=torch.bfloat16)
rs = torch.rand(M, device=dev) + 0.5
probe("plain (dgrad)", M, 4096, 14336, lambda: ops.gemm(x14, w_down, out=o4))
probe("residual (down / O proj, no fold)", M, 4096, 14336, lambda: ops.gemm(x14, w_down, residual=res, out=o4))
probe("residual + sumsq (fold)", M, 4096, 14336, lambda: ops.gemm_sumsq(x14, w_down, res))
probe("O proj residual + sumsq", M, 4096, 4096, lambda: ops.gemm_sumsq(x4, w_o, res))
probe("gate|up + SwiGLU fwd + row scale", M, 28672, 4096, lambda: ops.gemm_swiglu_fwd(x4, w_gu, row_scale=rs))
probe("down dgrad + SwiGLU bwd", M, 14336, 4096, lambda: ops.gemm_swiglu_bwd(x4, w_dT, gu))
for q in (False, True):
    probe("plain (dgrad)", M, 4096, 14336, lambda: ops.gemm(x14, w_down, out=o4), quarter=q)
    probe("O proj residual + sumsq", M, 4096, 4096, lambda: ops.gemm_sumsq(x4, w_o, res), quarter=q)
    probe("gate|up + SwiGLU fwd + row scale", M, 28672, 4096, lambda: ops.gemm_swiglu_fwd(x4, w_gu, row_scale=rs), quarter=q)
    probe("down dgrad + SwiGLU bwd", M, 14336, 4096, lambda: ops.gemm_swiglu_bwd(x4, w_dT, gu), quarter=q)
w_guT = torch.randn(4096, 28672, device=dev, dtype=torch.bfloat16) * 0.02
probe("gate|up dgrad (plain, K = 28672)", M, 4096, 28672, lambda: ops.gemm(gu, w_guT, out=o4))
