"""Parity bookkeeping shared by the GPU tests: every comparison against the oracle / a golden vector goes through `check`, which
prints the MEASURED error next to its bound and appends it to gpurun_out/parity_measured.jsonl (merged back from the GPU box), so
the bounds in the tests can be kept at measured-error x a small margin instead of generous guesses."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl")


def check(name, measured, bound):
    measured, bound = float(measured), float(bound)
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as fh:
            fh.write(json.dumps({"name": name, "measured": measured, "bound": bound}) + "\n")
    except OSError:
        pass
    print(f"[parity] {name}: measured {measured:.3e} (bound {bound:.1e})")
    assert measured <= bound, f"{name}: measured {measured:.4e} > bound {bound:.1e}"


def log(name, measured):
    """record a measured quantity (a yardstick, e.g. the bf16 CPU path's own deviation) without asserting anything"""
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as fh:
            fh.write(json.dumps({"name": name, "measured": float(measured), "bound": None}) + "\n")
    except OSError:
        pass
    print(f"[parity] {name}: measured {float(measured):.3e} (info)")


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)


def max_rel(got, want):
    """max |got - want| / max |want| (tensor-level relative error)."""
    return float((got.float() - want.float()).abs().max() / want.float().abs().max().clamp_min(1e-30))


def grad_err(got, want):
    """(1 - cosine, |norm ratio - 1|) of two flattened gradients."""
    import torch
    g, w = got.reshape(-1).double(), want.reshape(-1).double()
    cos = float(torch.dot(g, w) / (g.norm() * w.norm() + 1e-300))
    return 1.0 - cos, abs(float(g.norm() / (w.norm() + 1e-300)) - 1.0)


def scalar_grad_yardstick(O, ocfg, W, batch, names):
    """{name: (fp32-truth gradient, |bf16-CPU-path gradient - truth|)} for the scalar parameters `names` (logit scales): the yardstick for
    quantities that are sums of a few signed terms (|g| ~ 1e-6 by cancellation), where a relative bound proves nothing.  Both runs use the same
    bf16-rounded weights; the second is the reference-style CPU path (bf16 weights and activations, PyTorch's bf16 CPU ops)."""
    import torch
    BF = torch.bfloat16
    res = {}
    for mode in ("fp32", "bf16"):
        Wm = {k: (v.detach().to(BF) if (mode == "bf16" and v.dim() > 0) else v.detach().to(BF).float()) for k, v in W.items()}
        for k in names:
            Wm[k] = Wm[k].clone().float().requires_grad_(True)
        bm = {k: ((v.to(BF) if mode == "bf16" else v.to(BF).float()) if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v)
              for k, v in batch.items()}
        out = O.forward(Wm, bm, ocfg, need_logits=False)
        out["loss"].float().backward()
        res[mode] = {k: (0.0 if Wm[k].grad is None else float(Wm[k].grad.float().reshape(-1)[0])) for k in names}
    return {k: (res["fp32"][k], abs(res["bf16"][k] - res["fp32"][k])) for k in names}


def scalar_grad_bound(ref_abs, dev, old_rel):
    """5 % of the reference value + 1.5 x the bf16 CPU path's own deviation, never looser than the old relative bound"""
    return min(old_rel * abs(ref_abs), 0.05 * abs(ref_abs) + 1.5 * dev)
