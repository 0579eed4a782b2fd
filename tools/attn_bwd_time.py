"""Dev tool (gpurun): attention backward alone at the decoder shape (dQ + dK/dV kernels, RoPE^T fused like the step), 50 launches, three rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
B, Hq, Hkv, S, D = 8, 32, 8, 2048, 128
qkv = torch.randn(B, S, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
do = torch.randn(B, S, Hq, D, device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq = dqkv[..., :Hq * D].unflatten(-1, (Hq, D)); dk = dqkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); dv = dqkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
o, lse = ops.attn_fwd(q, k, v, True)
cos_t, sin_t = ops.rope_tables(S, D, 500000.0, qkv.device)
f = lambda: ops.attn_bwd(q, k, v, o, lse, do, True, dq=dq, dk=dk, dv=dv, rope=(cos_t, sin_t))
res = []
for r in range(3):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) / 50, 4))
print(os.environ.get("TAG", ""), res, "ms (dQ + dK/dV)", float(dqkv.float().abs().sum()))
