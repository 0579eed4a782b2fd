import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
M, N, K = 16384, 4096, 4096
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
for name, dbg in (("full", 0), ("no_glds", 1), ("no_dsread", 2), ("no_mfma", 4), ("no_glds_no_dsread", 3), ("only_barriers", 7), ("no_glds_no_mfma", 5), ("full_no_epilogue", 8), ("only_barriers_no_epi", 15)):
    ms = timeit(lambda: ops.gemm(a, w, out=out, epi=dbg << 8, force_generic=4))
    print(f"{name:22s} {ms*1e3:8.1f} us   ({2.0*M*N*K/ms/1e9:7.1f} TF/s equivalent)")
