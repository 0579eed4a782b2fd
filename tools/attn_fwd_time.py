"""Dev tool (gpurun): forward attention alone at the decoder shape, 100 back-to-back launches, three rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
B, Hq, Hkv, S, D = 8, 32, 8, 2048, 128
qkv = torch.randn(B, S, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
o = torch.empty(B, S, Hq, D, device="cuda", dtype=torch.bfloat16)
res = []
for r in range(3):
    for _ in range(10): ops.attn_fwd(q, k, v, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.attn_fwd(q, k, v, True)
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) / 100, 4))
print(os.environ.get("TAG", ""), res, "ms ->", round(4.0 * B * Hq * S * S * D / 2 / min(res) / 1e9), "TF/s")
