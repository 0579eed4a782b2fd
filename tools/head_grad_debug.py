"""Dev aid (gpurun): full-width heads, per-parameter gradient direction error vs the fp32 oracle, with the engine's weight-gradient GEMMs
and with an fp32 torch matmul monkey-patched over Engine._wgrad -> separates the wgrad kernels from everything upstream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fullwidth_gpu as T
from parity import grad_err
from visper_lm_amd.config import llama3_8b

cfg = llama3_8b(num_hidden_layers=2, vit_layers=4)
cfg.image_seg = dict(cfg.image_seg, seg_layer_indices="1")
cfg.image_depth = dict(cfg.image_depth, depth_layer_indices="2")
cfg.image_gen = dict(cfg.image_gen, img_layer_indices="2")
res = {}
for mode in ("engine", "torch_wgrad"):
    if mode == "torch_wgrad":
        from visper_lm_amd.engine import Engine

        def _torch_wgrad(self, x2d, dy2d, gview, accumulate=False):
            g2 = gview.view(dy2d.shape[1], x2d.shape[1])
            r = dy2d.float().t() @ x2d.float()
            g2.copy_(g2 + r if accumulate else r)
        Engine._wgrad = _torch_wgrad
    got, Wc, batch, tr = T._hip_step(cfg, 2, 128)
    res[mode] = got
ref = T._oracle_step(cfg, Wc, batch, tr, torch.float32, got["rows"])
for k, want in ref["grads"].items():
    if want is None or want.numel() == 1 or "depth" not in k:
        continue
    a = grad_err(res["engine"]["grads"][k], want); b = grad_err(res["torch_wgrad"]["grads"][k], want)
    print(f"{k:70s} engine 1-cos {a[0]:.2e} norm {a[1]:.2e} | torch-wgrad 1-cos {b[0]:.2e} norm {b[1]:.2e}")
