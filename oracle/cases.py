"""Seeded test cases shared by the golden generator, the oracle tests and the GPU parity tests
(TEST INFRASTRUCTURE).  Inputs are closed-form (oracle/weights.py) so fixtures store outputs only."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import weights as WT
from . import visper_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_batch(B, T, img_col, vocab_hi=1000, gen_dim=1024, depth_dim=1024, seg_dim=1536, with_targets=("gen", "depth", "seg")):
    """Same recipe as gen_golden.make_batch (ids uniform in [0,vocab_hi), one IMAGE token at img_col)."""
    ids = (WT.unit_uniform("input_ids", B * T).reshape(B, T) * 0.28867 + 0.5) * vocab_hi
    ids = torch.from_numpy(np.clip(ids, 0, vocab_hi - 1).astype(np.int64))
    ids[:, img_col] = O.IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, :img_col + 7] = O.IGNORE_INDEX
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids, dtype=torch.bool),
                 images=WT.tensor("images", (B, 3, 336, 336)))
    if "gen" in with_targets:
        batch["gen_target"] = WT.tensor("gen_target", (B, 1, gen_dim)); batch["gen_mask"] = torch.ones(B)
    if "depth" in with_targets:
        batch["depth_target"] = WT.tensor("depth_target", (B, 576, depth_dim)); batch["depth_mask"] = torch.ones(B)
    if "seg" in with_targets:
        batch["seg_target"] = WT.tensor("seg_target", (B, seg_dim, 24, 24)); batch["seg_mask"] = torch.ones(B)
    return batch


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tiny_llama_case(arch="llama"):
    """(cfg, W, batch, golden) for tests/golden/tiny_{llama,phi3}_e2e.npz."""
    g = load_golden(f"tiny_{arch}_e2e.npz")
    t = json.loads(str(g["cfg"]))
    extra = dict(rope_theta=10000.0, sliding_window=t.get("sliding_window")) if arch == "phi3" else {}
    cfg = O.make_config(arch=arch, **extra, vocab_size=t["vocab_size"], hidden_size=t["hidden_size"],
                        intermediate_size=t["intermediate_size"], num_hidden_layers=t["num_hidden_layers"],
                        num_attention_heads=t["num_attention_heads"], num_key_value_heads=t["num_key_value_heads"],
                        vit_hidden=t["vit_hidden"], vit_inter=t["vit_inter"], vit_layers=t["vit_layers"],
                        vit_heads=t["vit_heads"], aux_mode=t["aux_mode"], image_gen=t["image_gen"],
                        image_seg=t["image_seg"], image_depth=t["image_depth"])
    manifest = json.loads(str(g["manifest"]))
    W = {k: WT.param(k, s) for k, s in manifest.items()}
    B, T, col = json.loads(str(g["batch"])) if "batch" in g else (2, 59, 38)
    batch = make_batch(B, T, col)
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    return cfg, W, batch, g


def sub(t, n=4096):
    f = t.detach().float().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def tiny_ift_case():
    """(cfg, W, batch, golden) for tests/golden/tiny_llama_ift.npz: the reference's LlavaLlamaForCausalLM (IFT stage: no aux tasks,
    no task tokens; everything but the vision tower trainable)."""
    g = load_golden("tiny_llama_ift.npz")
    base, _, _, _ = tiny_llama_case()
    cfg = O.make_config(**{**vars(base), "aux_mode": "", "num_task_tokens": 0})
    W = {k: WT.param(k, s) for k, s in json.loads(str(g["manifest"])).items()}
    B, T, col = json.loads(str(g["batch"]))
    batch = {k: v for k, v in make_batch(B, T, col).items() if k in ("input_ids", "labels", "attention_mask", "images")}
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    return cfg, W, batch, g


def tiny_ift_tok_case():
    """(cfg, W, batch, golden) for tests/golden/tiny_llama_ift_tok.npz: the reference's LlavaLlamaForCausalLM built from a PT-stage
    config (num_task_tokens 8, aux_mode gen-depth-seg, task_token_format "emb"): raw (576, H) depth / seg rows + 8 gen rows behind the
    image (llava_arch.py:250-293), NTP loss only, everything but the vision tower trainable."""
    g = load_golden("tiny_llama_ift_tok.npz")
    base, _, _, _ = tiny_llama_case()
    t = json.loads(str(g["cfg"]))
    cfg = O.make_config(**{**vars(base), "aux_mode": t["aux_mode"], "num_task_tokens": t["num_task_tokens"], "image_depth": t["image_depth"],
                           "image_seg": t["image_seg"], "image_gen": t["image_gen"], "task_token_layout": "raw", "aux_heads": False})
    W = {k: WT.param(k, s) for k, s in json.loads(str(g["manifest"])).items()}
    B, T, col = json.loads(str(g["batch"]))
    batch = {k: v for k, v in make_batch(B, T, col).items() if k in ("input_ids", "labels", "attention_mask", "images")}
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    return cfg, W, batch, g


def tiny_nt0_case():
    """(cfg, W, batch, golden) for tests/golden/tiny_llama_nt0.npz: the PT step with num_task_tokens == 0 (plain Resampler heads)."""
    g = load_golden("tiny_llama_nt0.npz")
    base, _, _, _ = tiny_llama_case()
    cfg = O.make_config(**{**vars(base), "num_task_tokens": 0})
    W = {k: WT.param(k, s) for k, s in json.loads(str(g["manifest"])).items()}
    B, T, col = json.loads(str(g["batch"]))
    batch = make_batch(B, T, col)
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    return cfg, W, batch, g


def tiny_noid_case():
    """(cfg, W, batch, golden) for tests/golden/tiny_llama_noid.npz: the PT step with image_depth["use_intermediate_depth"] = False (no
    linear_1..3 in the depth head, loss on visual_feats, DPT on [feats[0]] * 4: base_ola_vlm.py:132,462-466, da_v2_head.py:437-455)."""
    g = load_golden("tiny_llama_noid.npz")
    base, _, _, _ = tiny_llama_case()
    t = json.loads(str(g["cfg"]))
    cfg = O.make_config(**{**vars(base), "image_depth": t["image_depth"]})
    W = {k: WT.param(k, s) for k, s in json.loads(str(g["manifest"])).items()}
    B, T, col = json.loads(str(g["batch"]))
    batch = make_batch(B, T, col)
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    return cfg, W, batch, g


def dinov2_weights(manifest):
    """Closed-form weights of the DINOv2 teacher fixture (same overrides as gen_golden.run_dinov2_teacher)."""
    W = {}
    for k, s in manifest.items():
        if k.endswith(".gamma"):
            W[k] = WT.tensor(k, s, 0.3, 1.0)
        elif k.endswith("cls_token") or k.endswith("pos_embed") or k.endswith("mask_token"):
            W[k] = WT.tensor(k, s, 0.2)
        else:
            W[k] = WT.param(k, s)
    return W


def swin_weights(manifest):
    """Closed-form weights of the Swin teacher fixture (same overrides as gen_golden.run_swin_teacher)."""
    return {k: (WT.tensor(k, s, 0.5) if k.endswith("relative_position_bias_table") else WT.param(k, s)) for k, s in manifest.items()}
