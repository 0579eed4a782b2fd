#!/bin/bash
# Round 6: does the store-data keep-alive (gemm.hip W4_KEEP2: 4-8 more live VGPRs in the epilogues) cost anything?  Product library (claim + keep-alive)
# vs variants/libvisper_claim_nokeep.so (claim, no keep-alive = round 5's epilogues), interleaved on one box: the perf-floor shapes alone, then the step.
root=$(pwd); V=$root/visper-lm_amd/variants
mb() { env "$@" python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from visper_lm_amd import ops
def ms(fn, n=8):
    fn(); fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return best
out = []
for (M, N, K) in [(16384, 4096, 4096), (16384, 14336, 4096), (16384, 4096, 14336), (16384, 6144, 4096)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); r = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    t = ms(lambda: ops.gemm(a, w, out=o)); t2 = ms(lambda: ops.gemm(a, w, residual=r, out=o))
    out.append(f"{M}x{N}x{K}: plain {2.0*M*N*K/t/1e9:.0f} residual {2.0*M*N*K/t2/1e9:.0f} TF/s")
print("   " + " | ".join(out))
PY
}
pt() { env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip()); print('   pt bench: ms/step', d['ms_per_step'], 'gemm frac', d['roofline']['frac'], d['roofline']['family']['frac'])"; }
for r in 1 2 3; do echo " keep-alive (product):"; mb VP_DUMMY=1; pt VP_DUMMY=1; echo " no keep-alive:"; mb VP_LIB_PATH=$V/libvisper_claim_nokeep.so; pt VP_LIB_PATH=$V/libvisper_claim_nokeep.so; done
