import sys, os
sys.path.insert(0, "/root/repo")
import torch
from visper_lm_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, K) in [(1024, 4096), (4096, 1024), (3072, 1024), (1024, 1024)]:
    row = {}
    for M, force in ((4616, 0), (4864, 8), (4864, 7), (4608, 8)):
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        row[(M, force)] = round(t(lambda: ops.gemm(a, w, out=o, force_generic=force)), 1)
    print((N, K), row, flush=True)
