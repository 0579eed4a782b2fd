"""Dev tool (gpurun): rmsnorm_bwd at the decoder shape over rotating buffers (cold HBM reads, like inside the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
M, H, N = 16384, 4096, 12
xs = [torch.randn(M, H, device="cuda", dtype=torch.bfloat16) for _ in range(N)]
dys = [torch.randn(M, H, device="cuda", dtype=torch.bfloat16) for _ in range(N)]
drs = [torch.randn(M, H, device="cuda", dtype=torch.bfloat16) for _ in range(N)]
w = torch.ones(H, device="cuda", dtype=torch.bfloat16)
rstd = torch.rand(M, device="cuda") + 0.5
for r in range(3):
    for i in range(N): ops.rmsnorm_bwd(dys[i], xs[i], w, rstd, dres=drs[i])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(4 * N): ops.rmsnorm_bwd(dys[i % N], xs[i % N], w, rstd, dres=drs[i % N])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (4 * N) * 1e3
    print(f"rmsnorm_bwd {us:.1f} us = {4 * M * H * 2 / us / 1e6:.2f} TB/s")
