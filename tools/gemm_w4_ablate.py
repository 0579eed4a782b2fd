"""Dev tool: sustained timing of the 4-wave GEMM kernel (force code 8) against the 8-phase kernel (7), 60 back-to-back launches each, HIP events
-> us per launch; alternating order, two passes.  (The ablation instantiations this tool timed while the K loop was being scheduled — force
codes 9 = no fragment reads, 10 = no reads + linear DMA source, 11 = no DMA; numbers in DESIGN.md section 4 — tripled the build time and are gone.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
for (M, N, K) in [(16384, 4096, 14336), (16384, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for rnd in range(2):
        res = {}
        for fg in (8, 7):
            for _ in range(10):
                ops.gemm(a, w, out=out, force_generic=fg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                ops.gemm(a, w, out=out, force_generic=fg)
            e1.record(); torch.cuda.synchronize()
            res[fg] = round(e0.elapsed_time(e1) / 60 * 1e3, 1)
        print(M, N, K, "us per launch:", res, flush=True)
