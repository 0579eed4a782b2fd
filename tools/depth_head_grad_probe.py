"""Dev tool (CPU only): WHERE does the 1 - cos = 3-5e-2 direction error of some depth-head weight gradients come from (the HIP step
and the reference-style bf16 CPU path both show it against fp32 truth, tests/test_fullwidth_gpu.py passes them through the adaptive yardstick)?

One full-width depth head (TaskTokenDepthHead, H = 4096, 576 queries, linear_1, smooth-L1 + contrastive loss) on a random layer state, weights and inputs
bf16-rounded once.  TRUTH = fp32 arithmetic.  Variants = the same fp32 arithmetic with the OUTPUT (forward) and the incoming GRADIENT (backward) of one class
of ops rounded to bf16 (what any bf16 implementation does between kernels): linear, layer_norm, softmax, gelu / relu, the loss-side normalisation; then all of
them (= the reference-style bf16 path's rounding points), and forward-only / backward-only rounding.  Printed: 1 - cos of every weight gradient against truth.

    python tools/depth_head_grad_probe.py [tokens]
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import visper_oracle as O          # noqa: E402  (dev tool: the oracle is the subject here, nothing is shipped)

BF = torch.bfloat16


class RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(BF).float() if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (g.to(BF).float() if ctx.bwd else g), None, None


class PatchedF:
    """torch.nn.functional with bf16 rounding behind chosen op classes (forward output and / or backward gradient)."""

    def __init__(self, which, fwd=True, bwd=True):
        self.which, self.fwd, self.bwd = set(which), fwd, bwd

    def _r(self, name, y):
        return RoundSTE.apply(y, self.fwd, self.bwd) if name in self.which else y

    def linear(self, *a, **k):
        return self._r("linear", F.linear(*a, **k))

    def layer_norm(self, *a, **k):
        return self._r("layer_norm", F.layer_norm(*a, **k))

    def gelu(self, *a, **k):
        return self._r("act", F.gelu(*a, **k))

    def relu(self, *a, **k):
        return self._r("act", F.relu(*a, **k))

    def __getattr__(self, n):
        return getattr(F, n)


def run(W, state, tgt, cfg, which, fwd=True, bwd=True):
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    O.F = PatchedF(which, fwd, bwd)
    old_softmax = torch.softmax
    if "softmax" in which:
        torch.softmax = lambda x, dim=-1: RoundSTE.apply(old_softmax(x, dim=dim), fwd, bwd)
    try:
        pred, _ = O.head_forward(state, "depth", 0, Wg, cfg)
        if "loss" in which:
            pred = RoundSTE.apply(pred, fwd, bwd)
        l3 = O.emb_loss(pred, torch.ones(pred.shape[0]), tgt, Wg["depth_logit_scale"], cfg.contrastive_loss_weight)
        l3[0].backward()
    finally:
        O.F = F
        torch.softmax = old_softmax
    return float(l3[0]), {k: v.grad for k, v in Wg.items() if v.grad is not None}


def main():
    n_text = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from visper_lm_amd.config import llama3_8b
    from visper_lm_amd.params import param_shapes, init_value
    c = llama3_8b(num_hidden_layers=2)
    c.image_depth = dict(c.image_depth, depth_layer_indices="2")
    cfg = O.make_config(**{k: v for k, v in c.to_dict().items() if k in vars(O.make_config())})
    gen = torch.Generator().manual_seed(3)
    W = {}
    for k, shp in param_shapes(c, vit_nested=True).items():
        if k.startswith("image_depth_heads.0.") or k in ("model.special_depth_tokens", "depth_logit_scale"):
            W[k] = init_value(k, shp, gen, torch.device("cpu"), BF if len(shp) else torch.float32).float()
    B, S = 2, 38 + 576 + 24 + n_text
    state = (torch.randn(B, S, c.hidden_size, generator=gen) * 1.0).to(BF).float()       # a layer state of unit scale
    tgt = torch.randn(B, 576, 1024, generator=gen).to(BF).float()
    loss0, g0 = run(W, state, tgt, cfg, [])
    variants = [("linear", ["linear"]), ("layer_norm", ["layer_norm"]), ("softmax", ["softmax"]), ("gelu/relu", ["act"]), ("loss input", ["loss"]),
                ("ALL (bf16 path's rounding points)", ["linear", "layer_norm", "softmax", "act", "loss"])]
    rows = []
    for name, which in variants:
        for tag, fw, bw in (("fwd+bwd", True, True), ("fwd only", True, False), ("bwd only", False, True)):
            if name != "ALL (bf16 path's rounding points)" and tag != "fwd+bwd":
                continue
            loss, g = run(W, state, tgt, cfg, which, fw, bw)
            errs = {k: 1.0 - float(F.cosine_similarity(g[k].flatten().double(), g0[k].flatten().double(), dim=0)) for k in g0 if g0[k].numel() > 1}
            rows.append((name, tag, abs(loss - loss0) / abs(loss0), errs))
    keys = sorted(rows[-1][3], key=lambda k: -rows[-3][3][k])[:8]
    print(f"depth head, H={c.hidden_size}, B={B}, S={S}: 1 - cos of weight gradients against fp32 truth (8 worst parameters under ALL)")
    print(f"{'rounded to bf16':38s} {'where':9s} {'loss rel':>9s}  " + "  ".join(k.replace('image_depth_heads.0.', '')[-26:].rjust(26) for k in keys))
    for name, tag, dl, errs in rows:
        print(f"{name:38s} {tag:9s} {dl:9.1e}  " + "  ".join(f"{errs[k]:26.2e}" for k in keys))


if __name__ == "__main__":
    main()


def per_linear():
    """Second table: round the output of ONE F.linear call at a time (call order inside head_forward: proj_in(latents), proj_in(x), to_q, to_kv, to_out,
    ff.1, ff.3, proj_out, linear_1.0, linear_1.2)."""
    names = ["proj_in(latents)", "proj_in(x)", "to_q", "to_kv", "to_out", "ff.1", "ff.3", "proj_out", "linear_1.0", "linear_1.2"]
    torch.manual_seed(0)
    from visper_lm_amd.config import llama3_8b
    from visper_lm_amd.params import param_shapes, init_value
    c = llama3_8b(num_hidden_layers=2)
    c.image_depth = dict(c.image_depth, depth_layer_indices="2")
    cfg = O.make_config(**{k: v for k, v in c.to_dict().items() if k in vars(O.make_config())})
    gen = torch.Generator().manual_seed(3)
    W = {}
    for k, shp in param_shapes(c, vit_nested=True).items():
        if k.startswith("image_depth_heads.0.") or k in ("model.special_depth_tokens", "depth_logit_scale"):
            W[k] = init_value(k, shp, gen, torch.device("cpu"), BF if len(shp) else torch.float32).float()
    B, S = 2, 38 + 576 + 24 + 128
    state = torch.randn(B, S, c.hidden_size, generator=gen).to(BF).float()
    tgt = torch.randn(B, 576, 1024, generator=gen).to(BF).float()
    _, g0 = run(W, state, tgt, cfg, [])
    key = "image_depth_heads.0.projector.proj_in.weight"
    print(f"\n1 - cos of {key} when ONE linear output is rounded to bf16 (forward):")
    for i, nm in enumerate(names):
        class OneLinear(PatchedF):
            def __init__(self):
                super().__init__([], True, False)
                self.n = 0

            def linear(self, *a, **k):
                y = F.linear(*a, **k)
                self.n += 1
                return RoundSTE.apply(y, True, False) if self.n - 1 == i else y
        Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
        O.F = OneLinear()
        try:
            pred, _ = O.head_forward(state, "depth", 0, Wg, cfg)
            O.emb_loss(pred, torch.ones(B), tgt, Wg["depth_logit_scale"], cfg.contrastive_loss_weight)[0].backward()
        finally:
            O.F = F
        e = 1.0 - float(F.cosine_similarity(Wg[key].grad.flatten().double(), g0[key].flatten().double(), dim=0))
        print(f"   {nm:18s} {e:.2e}   (output rms {float(pred.detach().pow(2).mean().sqrt()):.2f})")


if __name__ == "__main__" and os.environ.get("VP_PROBE_PER_LINEAR", "1") == "1":
    per_linear()
