"""IFT / VPT-stage API mirror: LlavaMetaModel / LlavaMetaForCausalLM (ola_vlm/model/llava_arch.py:36-486) and
LlavaLlamaForCausalLM / LlavaPhi3ForCausalLM (ola_vlm/model/language_model/llava_llama.py:39-175, llava_phi3.py) — the classes
scripts/train/finetune.sh trains (ola_vlm/train/train.py): next-token prediction only, the whole LLM + projector trainable,
vision tower frozen.  Same constructor / forward signature / attribute names / state-dict keys as the reference; the compute is
the engine's `train_llm` step (full weight gradients through the TN GEMM, per-layer gradient buckets, fused AdamW).

The nn.Parameters are VIEWS of the engine's flat store (EngineModule._get_engine), so the 8 B-parameter model exists once in HBM:
bf16 Parameters = the bf16 shadow the kernels read; the fp32 master and the AdamW moments are the optimizer's (DeepSpeed keeps the
same split in the reference: bf16 module weights + fp32 partitions, scripts/zero2.json)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from ..config import VisperConfig, phi3_mini
from .builders import ParamTree
from .language_model import EngineModule, _ModelOutput, _VisperStep
from .ola_arch import OlaLlavaMetaForCausalLM, OlaLlavaMetaModel


@dataclass
class CausalLMOutputWithPast(_ModelOutput):
    """transformers.modeling_outputs.CausalLMOutputWithPast (what llava_llama.py:108-119 returns)."""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    attentions: Optional[Tuple] = None


_IFT_DEFAULTS = dict(aux_mode="", num_task_tokens=0, train_llm=True, aux_heads=False)


class LlavaConfig(VisperConfig):
    """llava_llama.py:39-40 (`model_type = "llava_llama"`).  Defaults = finetune.sh on a plain LLaVA checkpoint: no task tokens.
    A PT-stage checkpoint's config carries aux_mode / num_task_tokens / task_token_format and is honoured (llava_arch.py:49-50)."""
    model_type = "llava_llama"

    def __init__(self, **kw):
        super().__init__(**{**_IFT_DEFAULTS, **kw, "train_llm": kw.get("train_llm", True), "aux_heads": False})


class LlavaPhi3Config(VisperConfig):
    model_type = "llava_phi3"

    def __init__(self, **kw):
        super().__init__(**{**phi3_mini().to_dict(), **_IFT_DEFAULTS, **kw, "train_llm": kw.get("train_llm", True), "aux_heads": False})


class LlavaMetaModel(OlaLlavaMetaModel):
    """llava_arch.py:36-130: vision tower, mm_projector, special_{depth,seg,gen}_tokens — the same surface as OlaLlavaMetaModel."""


class LlavaMetaForCausalLM(OlaLlavaMetaForCausalLM):
    """llava_arch.py:210-486: encode_images (:295-298), prepare_inputs_labels_for_multimodal (:300-486), the token properties.

    append_special_tokens (:240-293) differs from the PT stage in ONE way: with task_token_format == "emb" it splices the raw
    (576, H) depth / seg parameters (all rows), while "expand_emb" mean-pools them to num_task_tokens rows like ola_arch.py.  The
    engine implements the pooled layout; the "emb" / "text" layouts with num_task_tokens > 0 are refused loudly."""

    def _check_task_token_format(self):
        cfg = self.config
        if cfg.num_task_tokens > 0 and cfg.token_order and getattr(cfg, "task_token_format", "emb") != "expand_emb":
            raise NotImplementedError(
                f"LlavaMetaForCausalLM with num_task_tokens={cfg.num_task_tokens} and task_token_format="
                f"{cfg.task_token_format!r}: only 'expand_emb' (mean-pooled rows, llava_arch.py:252-254) or num_task_tokens == 0 "
                "are implemented on the MI355X path")


class LlavaLlamaModel(LlavaMetaModel, ParamTree):
    config_class = LlavaConfig


class LlavaPhi3Model(LlavaMetaModel, ParamTree):
    config_class = LlavaPhi3Config


class _LlavaCausalLMBase(LlavaMetaForCausalLM, EngineModule):
    model_cls = LlavaLlamaModel

    def __init__(self, config, device="cuda", dtype=torch.bfloat16, init="random", seed=0):
        config.aux_heads = False
        EngineModule.__init__(self, config, device=device, dtype=dtype, init=init, seed=seed)
        self._check_task_token_format()

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None, image_sizes=None,
                return_dict=None, **kwargs):
        """llava_llama.py:73-119 signature -> CausalLMOutputWithPast.  Extra kwarg: output_logits=True materialises `logits`
        (the reference always returns the full fp32 logits; the fused lm_head + CE path never builds them unless asked)."""
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("the MI355X path covers the training forward (input_ids + images); generation is out of scope")
        if images is None:
            raise NotImplementedError("text-only batches without an `images` tensor are not wired (the reference feeds a dummy image)")
        self._sync_trainable()
        eng = self._get_engine()
        batch = dict(input_ids=input_ids, attention_mask=attention_mask, labels=labels, images=images.to(eng.dev))
        eng.keep_logits = bool(kwargs.get("output_logits", False))
        if labels is not None:
            loss = _VisperStep.apply(self, batch, *self._trainable_params)
        else:
            self._last = eng.train_step(batch, compute_grads=False)
            loss = None
        out = self._last
        return CausalLMOutputWithPast(loss=loss, logits=out.get("logits"), hidden_states=(out["hidden"],))


class LlavaLlamaForCausalLM(_LlavaCausalLMBase):
    config_class = LlavaConfig
    model_cls = LlavaLlamaModel


class LlavaPhi3ForCausalLM(_LlavaCausalLMBase):
    config_class = LlavaPhi3Config
    model_cls = LlavaPhi3Model
