"""build_vision_tower / build_vision_projector / CLIPVisionTower mirrors
(ola_vlm/model/multimodal_encoder/builder.py, clip_encoder.py:7-90; multimodal_projector/builder.py:47-65).
They are parameter containers with the reference's state-dict layout; forward runs on the HIP engine."""
from __future__ import annotations

import re

import torch
import torch.nn as nn


class ParamTree(nn.Module):
    """nn.Module whose nested children/parameters are created from dotted state-dict names."""

    def add(self, dotted, shape, device, dtype, requires_grad=False):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, ParamTree())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, device=device, dtype=dtype), requires_grad=requires_grad))

    def __getitem__(self, i):          # nn.Sequential-style indexing (mm_projector[0], heads[i])
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


class CLIPVisionTower(ParamTree):
    """clip_encoder.py:7-90.  `forward(images)` = frozen tower -> hidden_states[select_layer] (CLS dropped)."""

    def __init__(self, vision_tower="openai/clip-vit-large-patch14-336", args=None, delay_load=False):
        super().__init__()
        self.is_loaded = True
        self.vision_tower_name = vision_tower
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.__dict__["_owner"] = None

    @torch.no_grad()
    def forward(self, images):
        eng = self._owner._get_engine()
        B = images.shape[0]
        feats = eng.vit_forward(images.to(eng.dev))
        return feats.view(B, -1, feats.shape[-1]).to(images.dtype)

    @property
    def hidden_size(self):
        return self._owner.config.vit_hidden

    @property
    def num_patches_per_side(self):
        c = self._owner.config
        return c.vit_image // c.vit_patch

    @property
    def num_patches(self):
        return self.num_patches_per_side ** 2


class CLIPConvNextVisionTower(CLIPVisionTower):
    """clip_convnext_encoder.py:61-205 (timm ConvNeXt trunk via open_clip): stem -> stages -> norm_pre -> (B, 576, C)."""

    @property
    def hidden_size(self):
        return self._owner.config.cnx_dims[-1]

    @property
    def num_patches_per_side(self):
        return self._owner.config.cnx_image // 32


def build_vision_tower(vision_tower_cfg, **kwargs):
    name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if name is not None and "convnext" in str(name).lower():
        return CLIPConvNextVisionTower(name, args=vision_tower_cfg, **kwargs)
    return CLIPVisionTower(name, args=vision_tower_cfg, **kwargs)


def build_vision_projector(config, delay_load=False, **kwargs):
    """multimodal_projector/builder.py:47-65 — only the layouts the PT scripts use."""
    t = getattr(config, "mm_projector_type", "linear")
    m = re.match(r"^mlp(\d+)x_gelu$", t)
    if not (m and int(m.group(1)) == 2):
        raise NotImplementedError(f"mm_projector_type={t!r}: the MI355X path implements mlp2x_gelu (pretrain.sh)")
    return ParamTree()
