"""Cost of the batched DINOv2-L depth teacher at the training batch (dev tool; gpurun)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd.teachers import DinoV2DepthTeacher

t = DinoV2DepthTeacher()
g = torch.Generator(device="cuda").manual_seed(0)
W = {k: (torch.randn(s, device="cuda", generator=g) * (0.02 if len(s) > 1 else 1.0)) for k, s in DinoV2DepthTeacher.shapes().items()}
for k in W:
    if k.endswith(".gamma") or k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight"):
        W[k] = torch.ones_like(W[k])
t.load_weights(W)
x = torch.randn(8, 3, 336, 336, device="cuda")
for _ in range(2):
    y = t.forward(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    y = t.forward(x)
e1.record(); torch.cuda.synchronize()
print(f"DINOv2-L depth teacher, B=8, 336 px: {e0.elapsed_time(e1) / 5:.2f} ms per batch, out {tuple(y.shape)}, finite {bool(torch.isfinite(y.float()).all())}")
