"""OlaLlavaMetaModel / OlaLlavaMetaForCausalLM mirrors (ola_vlm/model/ola_arch.py:36-444)."""
from __future__ import annotations

import torch


class OlaLlavaMetaModel:
    """Mixin for the `.model` object: vision tower, projector, special task tokens (ola_arch.py:36-144)."""

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        return vt[0] if isinstance(vt, list) else vt

    def get_special_tokens(self):
        return (getattr(self, "special_depth_tokens", None), getattr(self, "special_seg_tokens", None),
                getattr(self, "special_gen_tokens", None))

    def initialize_special_tokens(self, config):        # ola_arch.py:67-94 (parameters are created with the manifest)
        self.num_task_tokens = config.num_task_tokens
        self.task_token_format = getattr(config, "task_token_format", "emb")
        self.is_sample_tokens = getattr(config, "sample_tokens", False)
        self.aux_tokens = config.aux_mode
        self.token_order = config.aux_mode.split("-")

    def initialize_vision_modules(self, model_args, fsdp=None):     # ola_arch.py:96-144
        cfg = self.config
        cfg.mm_vision_tower = getattr(model_args, "vision_tower", cfg.mm_vision_tower)
        cfg.use_mm_proj = True
        cfg.mm_projector_type = getattr(model_args, "mm_projector_type", "mlp2x_gelu")
        cfg.mm_vision_select_layer = getattr(model_args, "mm_vision_select_layer", -2)
        cfg.mm_vision_select_feature = getattr(model_args, "mm_vision_select_feature", "patch")
        for p in self.mm_projector.parameters():
            p.requires_grad = True


class OlaLlavaMetaForCausalLM:
    """Mixin for the CausalLM object (ola_arch.py:179-444)."""

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def encode_images(self, images):                     # ola_arch.py:187-190
        eng = self._get_engine()
        from .. import ops
        feats = eng.vit_forward(images.to(eng.dev))
        ps = eng.ps
        z1 = ops.gemm(feats, ps.w("model.mm_projector.0.weight"), bias=ps.w("model.mm_projector.0.bias"))
        a1 = ops.act_fwd(z1, ops.EPI_GELU)
        out = ops.gemm(a1, ps.w("model.mm_projector.2.weight"), bias=ps.w("model.mm_projector.2.bias"))
        return out.view(images.shape[0], -1, out.shape[-1]).to(images.dtype)

    depth_tokens = property(lambda self: self.get_model().get_special_tokens()[0])
    seg_tokens = property(lambda self: self.get_model().get_special_tokens()[1])
    gen_tokens = property(lambda self: self.get_model().get_special_tokens()[2])
    num_task_tokens = property(lambda self: self.get_model().num_task_tokens)
    task_token_format = property(lambda self: self.get_model().task_token_format)
    aux_tokens = property(lambda self: self.get_model().aux_tokens)
    token_order = property(lambda self: self.get_model().token_order)
    is_sample_tokens = property(lambda self: self.get_model().is_sample_tokens)

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                             image_sizes=None):
        """ola_arch.py:256-444 -> (None, position_ids, attention_mask, past_key_values, inputs_embeds, labels).
        The splice itself is a HIP row gather driven by the host-built index plan (engine.build_plan)."""
        if self.get_vision_tower() is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        eng = self._get_engine()
        self._sync_trainable()
        embeds, plan = eng.splice_forward(input_ids, attention_mask, labels, images)
        S = plan["S"]
        new_labels = None if labels is None else plan["labels"].to(input_ids.device)
        am = None if attention_mask is None else plan["attention_mask"].to(device=input_ids.device, dtype=attention_mask.dtype)
        pid = None if position_ids is None else plan["position_ids"].to(input_ids.device)
        return None, pid, am, past_key_values, embeds, new_labels

    def initialize_vision_tokenizer(self, model_args, tokenizer):      # ola_arch.py:446-489: tokenizer plumbing only
        if getattr(model_args, "mm_use_im_start_end", False) or getattr(model_args, "mm_use_im_patch_token", False):
            raise NotImplementedError("im_start/end and im_patch tokens are not used by the PT scripts (pretrain.sh)")
