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

# ---- the other two teachers at the training batch
from visper_lm_amd.teachers import ClipImageEmbedTeacher, SwinSegTeacher
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection, SwinConfig, SwinBackbone


def rand_like_state(model, pre):
    gg = torch.Generator(device="cuda").manual_seed(1)
    out = {}
    for k, v in model.state_dict().items():
        if "position_ids" in k:
            continue
        if v.dim() == 1 and ("norm" in k.lower() or "layernorm" in k.lower()) and k.endswith("weight"):
            out[pre + k] = torch.ones(v.shape, device="cuda")
        else:
            out[pre + k] = torch.randn(v.shape, device="cuda", generator=gg) * (0.02 if v.dim() > 1 else 0.01)
    return out


def bench(name, t, x):
    for _ in range(2):
        y = t.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        y = t.forward(x)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 3:.2f} ms per batch, out {tuple(y.shape)}, finite {bool(torch.isfinite(y.float()).all())}")


with torch.device("meta"):
    vh = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16,
                                                        image_size=224, patch_size=14, projection_dim=1024, hidden_act="gelu"))
    sw = SwinBackbone(SwinConfig(image_size=768, patch_size=4, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=12,
                                 out_features=["stage4"]))
tc = ClipImageEmbedTeacher()
tc.load_weights(rand_like_state(vh, "pipe.image_encoder."))
bench("CLIP ViT-H/14 image-embed teacher, B=8, 224 px", tc, torch.randn(8, 3, 224, 224, device="cuda"))
ts = SwinSegTeacher()
ts.load_weights(rand_like_state(sw, "oneformer.model.pixel_level_module.encoder."))
bench("Swin-L segmentation teacher, B=8, 768 px", ts, torch.randn(8, 3, 768, 768, device="cuda"))
