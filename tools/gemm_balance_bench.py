"""XCD speed balancing of the persistent GEMM: time with / without, and check bit-identity (dev tool; run through gpurun)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops

dev = "cuda"
sp = ops.calibrate_xcd_balance(dev, force=True)
print("calibrated XCD speeds:", sp)
shapes = [(16384, 4096, 4096), (16384, 6144, 4096), (16384, 4096, 14336), (16384, 28672, 4096)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    o0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    o1 = torch.empty_like(o0)
    res = {}
    for rnd in range(3):
        for name, speeds, out in (("off", None, o0), ("on", sp, o1)):
            ops.set_xcd_speeds(speeds)
            for _ in range(10):
                ops.gemm(a, w, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.gemm(a, w, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 30 * 1e3)
    fl = 2.0 * M * N * K
    print(f"{(M, N, K)}: off {[round(x, 1) for x in res['off']]} us  on {[round(x, 1) for x in res['on']]} us  "
          f"-> {fl / min(res['off']) / 1e6:.0f} vs {fl / min(res['on']) / 1e6:.0f} TF/s  bit-identical: {torch.equal(o0, o1)}")

# per-XCD block end times with / without balancing (stamps on: 0x10000; keep balancing: 0x100000)
import ctypes as C, numpy as np
from visper_lm_amd import _lib
M, N, K = 16384, 4096, 14336
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for name, speeds in (("off", None), ("on", sp), ("off", None), ("on", sp)):
    ops.set_xcd_speeds(speeds)
    for _ in range(20):
        ops.gemm(a, w, out=o)
    _lib.call("vp_debug_gemm_flags", 0x10000 | 0x100000)
    ops.gemm(a, w, out=o); torch.cuda.synchronize()
    _lib.call("vp_debug_gemm_flags", 0)
    buf = (C.c_long * 2048)(); _lib.call("vp_debug_stamps", buf)
    st = np.array(buf[:], dtype=np.int64).reshape(256, 8)
    end = (st[:, 5] - st[:, 0].min()) / 100.0
    print(name, "end us by xcd:", [round(float(end[x::8].mean()), 1) for x in range(8)], "max", round(float(end.max()), 1), "mean", round(float(end.mean()), 1))
