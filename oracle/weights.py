"""Closed-form tensors for parity tests (TEST INFRASTRUCTURE).

Both the golden generator (which fills the *reference* model's parameters by state-dict name)
and the tests (which build the flat weight dict for the oracle / the HIP engine) regenerate
identical values from (name, shape) alone, so no weights are stored in fixtures.

value[i] = amp * u(i),  u = murmur3-fmix64(i + crc32(name) * golden) mapped to uniform[-sqrt3, sqrt3)
(unit variance, full-rank — a sin() recipe would make every matrix rank 2).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch

_M1 = np.uint64(0xFF51AFD7ED558CCD)
_M2 = np.uint64(0xC4CEB9FE1A85EC53)
_G = np.uint64(0x9E3779B97F4A7C15)


def unit_uniform(name: str, n: int) -> np.ndarray:
    """n unit-variance pseudo-random float64 values, a pure function of (name, index)."""
    with np.errstate(over="ignore"):
        k = np.arange(n, dtype=np.uint64) + np.uint64(zlib.crc32(name.encode())) * _G
        k ^= k >> np.uint64(33)
        k *= _M1
        k ^= k >> np.uint64(33)
        k *= _M2
        k ^= k >> np.uint64(33)
    u = (k >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (u * 2.0 - 1.0) * np.sqrt(3.0)


def tensor(name: str, shape, amp: float = 1.0, mean: float = 0.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = unit_uniform(name, n) * amp + mean
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape)).to(dtype)


def param(name: str, shape) -> torch.Tensor:
    """Init rule by parameter name/shape (fp32)."""
    shape = tuple(shape)
    last = name.rsplit(".", 1)[-1]
    if name.endswith("logit_scale"):
        return torch.tensor(2.0)                                  # base_ola_vlm.py:113
    if "special_" in name and name.endswith("_tokens"):
        return tensor(name, shape, 1.0)                           # ~N(0,1): ola_arch.py:80-94
    is_norm = any(s in name for s in ("layernorm", "layer_norm", "layrnorm", "norm1", "norm2", "norm_out",
                                      ".norm.", "layers.0.1.0.")) or name.endswith("model.norm.weight")
    if len(shape) == 1:
        if last == "bias":
            return tensor(name, shape, 0.02)
        if is_norm or last == "weight":
            return tensor(name, shape, 0.1, 1.0)
        return tensor(name, shape, 0.05)                          # class_embedding
    if "embed_tokens" in name or "position_embedding" in name:
        return tensor(name, shape, 0.05)
    fan_in = int(np.prod(shape[1:]))
    if name.startswith("da_v2_head."):
        # ~25 ReLU convs deep: He-style gain keeps the signal alive; ConvTranspose2d weights are [in, out, k, k] and
        # each output pixel sees exactly `in` taps
        if "resize_layers.0." in name or "resize_layers.1." in name:
            fan_in = shape[0]
        return tensor(name, shape, 1.4 / np.sqrt(fan_in))
    return tensor(name, shape, 0.8 / np.sqrt(fan_in))


def fill_state(shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, torch.Tensor]:
    return {k: param(k, s) for k, s in shapes.items()}
