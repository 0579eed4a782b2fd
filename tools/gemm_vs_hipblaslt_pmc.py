"""Dev tool (run via tools/gemm_vs_hipblaslt_pmc.sh under rocprofv3): 40 back-to-back launches per kernel and shape — our default GEMM and
hipBLASLt through torch.matmul (comparison baseline only) — so the package sits at its power cap like inside the step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
SHAPES = [(16384, 4096, 4096), (16384, 6144, 4096), (16384, 28672, 4096), (16384, 4096, 14336)]
n = int(os.environ.get("REPS", "40"))
FORCE = int(os.environ.get("FORCE", "0"))          # 0 = the default kernel, 8 = the 4-wave kernel
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for which in ("ours", "hipblaslt", "ours", "hipblaslt"):
        for _ in range(n):
            if which == "ours":
                ops.gemm(a, w, out=out, force_generic=FORCE)
            else:
                torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()
