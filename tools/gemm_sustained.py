"""Sustained (40 back-to-back launches) TFLOP/s of the auto-routed GEMM on the decoder shapes (dev tool for build-variant A/Bs; gpurun)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
shapes = [(16384, 4096, 4096), (16384, 6144, 4096), (16384, 4096, 14336), (16384, 28672, 4096)]
res = []
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    r = torch.randn(M, N, device="cuda", dtype=torch.bfloat16); o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10): ops.gemm(a, w, residual=r, out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): ops.gemm(a, w, residual=r, out=o)
    e1.record(); torch.cuda.synchronize()
    res.append(round(2.0 * M * N * K * 40 / e0.elapsed_time(e1) / 1e9))
print("TF/s", dict(zip(["4096x4096", "6144x4096", "4096x14336", "28672x4096"], res)), flush=True)
