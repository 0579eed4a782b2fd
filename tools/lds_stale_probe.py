"""Does stale LDS content (NaN tiles left by an earlier kernel) leak into the valid rows of an M-tail launch of the general 4-wave variant?  (gpurun)"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from visper_lm_amd import ops
BF = torch.bfloat16
nan_a = torch.full((65536, 1024), float("nan"), device="cuda", dtype=BF)
nan_w = torch.full((4096, 1024), float("nan"), device="cuda", dtype=BF)
for (M, N, K) in [(4616, 3072, 1024), (4616, 1024, 4096), (4616, 4096, 1024), (4616, 1024, 1024), (4616, 1024, 640), (4608, 4096, 1024)]:
    a = torch.randn(M, K, device="cuda", dtype=BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF); b = torch.randn(N, device="cuda", dtype=BF)
    r = torch.randn(M, N, device="cuda", dtype=BF)
    ref = ops.gemm(a, w, bias=b, residual=r, force_generic=7)
    bad = 0
    for rep in range(5):
        ops.gemm(nan_a, nan_w, force_generic=8)                 # every CU's LDS buffers now hold NaN tiles
        got = ops.gemm(a, w, bias=b, residual=r, force_generic=14)
        bad += int(torch.isnan(got.float()).sum())
        assert torch.equal(got, ref) or bad, "mismatch without NaN?"
    print((M, N, K), "NaN elements in valid rows over 5 runs:", bad, flush=True)
