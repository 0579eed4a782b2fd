"""Host side of the optimizer step: learning-rate schedule, parameter groups and gradient clipping exactly as
the reference's trainer sets them up, driving the fused `vp_adamw` kernel on contiguous runs of the flat parameter buffer.

Reference: HF `adamw_torch` (ola_vlm_train.py:124; betas (0.9, 0.999), eps 1e-8), cosine schedule with `warmup_ratio 0.03`
(scripts/train/pretrain.sh:45-48 -> transformers.get_cosine_schedule_with_warmup), parameter groups of
`LLaVATrainer.create_optimizer` (ola_vlm/train/llava_trainer.py:890-995): weight decay on everything except biases and LayerNorm
parameters, optional `mm_projector_lr` for the projector.  `scripts/zero2.json` has no `gradient_clipping` key, so clipping is off
in the reference run; `max_grad_norm` is offered with `torch.nn.utils.clip_grad_norm_` semantics (global L2 norm, eps 1e-6)."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Tuple


def warmup_steps(total_steps: int, warmup_ratio: float = 0.03) -> int:
    """HF TrainingArguments.get_warmup_steps: ceil(total * ratio)."""
    return int(math.ceil(total_steps * warmup_ratio))


def cosine_with_warmup(step: int, total_steps: int, n_warmup: int, num_cycles: float = 0.5) -> float:
    """LR multiplier of transformers.get_cosine_schedule_with_warmup at optimizer step `step` (0-based: the value used BY step
    `step`, i.e. LambdaLR after `step` scheduler.step() calls)."""
    if step < n_warmup:
        return float(step) / float(max(1, n_warmup))
    progress = float(step - n_warmup) / float(max(1, total_steps - n_warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


_NORM_MARKERS = (".norm1.", ".norm2.", ".norm_out.", "layernorm", "layer_norm", ".norm.")


def is_no_decay(name: str) -> bool:
    """True for parameters the reference puts in the weight_decay = 0 groups: biases and everything inside an nn.LayerNorm /
    RMSNorm module (transformers.trainer_pt_utils.get_parameter_names(model, ALL_LAYERNORM_LAYERS) + the `"bias" not in name`
    filter, llava_trainer.py:903-904).  In the PT trainable set the LayerNorms are the resampler's norm1 / norm2 / norm_out and the
    first module of each FeedForward (`layers.<d>.1.0`, resampler.py:9-16)."""
    if "bias" in name:
        return True
    n = "." + name
    if any(m in n for m in _NORM_MARKERS):
        return True
    parts = name.split(".")
    # FeedForward = Sequential(LayerNorm, Linear, GELU, Linear): "...layers.<d>.1.0.weight"
    return len(parts) >= 5 and parts[-5] == "layers" and parts[-3] == "1" and parts[-2] == "0"


def param_groups(names: Iterable[str], weight_decay: float = 0.0, mm_projector_lr: Optional[float] = None) -> Dict[str, Tuple[Optional[float], float]]:
    """name -> (group learning rate or None for the trainer's base lr, weight decay)."""
    out = {}
    for n in names:
        lr = mm_projector_lr if (mm_projector_lr is not None and "mm_projector" in n) else None
        out[n] = (lr, 0.0 if is_no_decay(n) else float(weight_decay))
    return out


def runs(index: "Dict[str, Tuple[int, int, tuple]]", groups: Dict[str, Tuple[Optional[float], float]]) -> List[Tuple[int, int, Optional[float], float]]:
    """Merge the flat buffer's per-parameter slots (ParamStore.index: name -> (offset, numel, shape), slots padded to 64) into
    maximal contiguous runs with one (lr, weight_decay) each: one `vp_adamw` launch per run."""
    items = sorted(((off, n, groups[name]) for name, (off, n, _) in index.items()), key=lambda t: t[0])
    out: List[Tuple[int, int, Optional[float], float]] = []
    for i, (off, n, (lr, wd)) in enumerate(items):
        end = items[i + 1][0] if i + 1 < len(items) else off + (n + 63) // 64 * 64
        if out and out[-1][2] == lr and out[-1][3] == wd and out[-1][1] == off:
            out[-1] = (out[-1][0], end, lr, wd)
        else:
            out.append((off, end, lr, wd))
    return out


def clip_coefficient(total_norm: float, max_norm: float, eps: float = 1e-6) -> float:
    """torch.nn.utils.clip_grad_norm_: grads *= min(1, max_norm / (total_norm + eps))."""
    return min(1.0, float(max_norm) / (float(total_norm) + eps))
