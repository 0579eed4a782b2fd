"""Host logic of the optimizer step: LR schedule, parameter groups, run merging, clipping coefficient — against
the libraries the reference's trainer uses (transformers' scheduler and parameter-name rule, torch's clip_grad_norm_)."""
import json
import math

import numpy as np
import pytest
import torch

from visper_lm_amd import optim


def test_cosine_schedule_matches_transformers():
    from transformers import get_cosine_schedule_with_warmup
    for total, ratio in ((100, 0.03), (2181, 0.03), (10, 0.5), (7, 0.0)):
        nw = optim.warmup_steps(total, ratio)
        assert nw == math.ceil(total * ratio)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=1.0)
        sch = get_cosine_schedule_with_warmup(opt, nw, total)
        for step in range(total):
            assert abs(sch.get_last_lr()[0] - optim.cosine_with_warmup(step, total, nw)) < 1e-12, (total, ratio, step)
            opt.step(); sch.step()


def test_param_groups_follow_the_reference_rule():
    """llava_trainer.py:903-904: decay = get_parameter_names(model, ALL_LAYERNORM_LAYERS) minus names containing "bias"."""
    from oracle import cases
    _, _, _, g = cases.tiny_llama_case()
    trainable = json.loads(str(g["trainable"]))
    groups = optim.param_groups(trainable, weight_decay=0.1, mm_projector_lr=2e-5)
    for n in trainable:
        lr, wd = groups[n]
        ln = any(k in n for k in ("norm1", "norm2", "norm_out")) or n.split(".")[-3:-1] == ["1", "0"]
        assert (wd == 0.0) == ("bias" in n or ln), n
        assert (lr == 2e-5) == ("mm_projector" in n), n
    # top-level nn.Parameters (task tokens, logit scales) are decayed like any weight (they are in model._parameters)
    assert groups["model.special_seg_tokens"][1] == 0.1 and groups["seg_logit_scale"][1] == 0.1
    assert groups["image_seg_heads.0.projector.layers.0.1.0.weight"][1] == 0.0      # FeedForward's LayerNorm
    assert groups["image_seg_heads.0.projector.layers.0.1.1.weight"][1] == 0.1      # FeedForward's first Linear


def test_runs_cover_the_flat_buffer_once():
    index, off = {}, 0
    names = ["a.weight", "a.bias", "b.norm1.weight", "b.norm1.bias", "model.mm_projector.0.weight", "model.mm_projector.0.bias", "c.weight"]
    for i, n in enumerate(names):
        numel = 100 + 37 * i
        index[n] = (off, numel, (numel,))
        off += (numel + 63) // 64 * 64
    one = optim.runs(index, optim.param_groups(names, 0.0, None))
    assert one == [(0, off, None, 0.0)]                                              # reference configuration: one launch
    many = optim.runs(index, optim.param_groups(names, 0.1, 2e-5))
    assert many[0][0] == 0 and many[-1][1] == off and all(a[1] == b[0] for a, b in zip(many, many[1:]))
    assert [(r[2], r[3]) for r in many] == [(None, 0.1), (None, 0.0), (2e-5, 0.1), (2e-5, 0.0), (None, 0.1)]


def test_clip_coefficient_matches_torch():
    g = torch.randn(1000) * 3
    for mx in (0.5, 1.0, 1e3):
        p = torch.nn.Parameter(torch.zeros(1000)); p.grad = g.clone()
        tot = torch.nn.utils.clip_grad_norm_([p], mx)
        c = optim.clip_coefficient(float(g.norm()), mx)
        assert np.allclose((g * c).numpy(), p.grad.numpy(), rtol=1e-6) and abs(float(tot) - float(g.norm())) < 1e-4
