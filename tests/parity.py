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
