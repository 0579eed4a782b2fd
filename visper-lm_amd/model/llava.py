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
from .language_model import EngineModule, _ModelOutput, _VisperStep, _run_engine
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


def _ift_kwargs(kw):
    """IFT-stage trainability is a property of the CLASS, not of a stored config: the reference's train.py makes the LLM trainable
    whatever the checkpoint's config.json says (train.py:1045-1068; a PT-stage config.json always carries the PT run's
    `train_llm: false` because save_pretrained dumps config.to_dict()).  So `train_llm` / `aux_heads` / `model_type` found in `kw`
    are ignored; the only way to freeze the LLM in these classes is the explicit `freeze_llm=True` (constructor or from_pretrained
    override; the analogue of train.py's `freeze_backbone`)."""
    kw = dict(kw)
    if kw.get("train_llm") is False:
        # an explicit constructor argument (the loaders drop a STORED train_llm / aux_heads before they get here: STORED_KEYS_IGNORED):
        # say loudly that the value is not honoured
        import warnings
        warnings.warn("train_llm=False is ignored by the IFT-stage classes (a PT checkpoint's config always carries it); pass freeze_llm=True "
                      "to keep the LLM frozen", stacklevel=3)
    freeze = bool(kw.pop("freeze_llm", False))
    for k in ("train_llm", "aux_heads", "model_type"):
        kw.pop(k, None)
    return {**_IFT_DEFAULTS, **kw, "train_llm": not freeze, "aux_heads": False, "freeze_llm": freeze}


class LlavaConfig(VisperConfig):
    """llava_llama.py:39-40 (`model_type = "llava_llama"`).  Defaults = finetune.sh on a plain LLaVA checkpoint: no task tokens.
    A PT-stage checkpoint's config carries aux_mode / num_task_tokens / task_token_format and is honoured (llava_arch.py:49-50,
    67-94): the task-token rows are spliced in the layout `task_token_format` names (LlavaMetaForCausalLM below)."""
    model_type = "llava_llama"
    STORED_KEYS_IGNORED = ("train_llm", "aux_heads")       # dropped from a loaded config.json (EngineModule.from_pretrained, hf_auto)

    def __init__(self, **kw):
        super().__init__(**_ift_kwargs(kw))


class LlavaPhi3Config(VisperConfig):
    model_type = "llava_phi3"
    STORED_KEYS_IGNORED = ("train_llm", "aux_heads")

    def __init__(self, **kw):
        base = phi3_mini().to_dict()
        base.pop("model_type", None)                     # phi3_mini() tags its instance "ola_phi3": would shadow the class attribute
        for k in self.STORED_KEYS_IGNORED:               # (the preset's PT-stage defaults, not a caller's choice)
            base.pop(k, None)
        super().__init__(**_ift_kwargs({**base, **kw}))


class LlavaMetaModel(OlaLlavaMetaModel):
    """llava_arch.py:36-130: vision tower, mm_projector, special_{depth,seg,gen}_tokens — the same surface as OlaLlavaMetaModel."""


class LlavaMetaForCausalLM(OlaLlavaMetaForCausalLM):
    """llava_arch.py:210-486: encode_images (:295-298), prepare_inputs_labels_for_multimodal (:300-486), the token properties.

    append_special_tokens (:250-293) differs from the PT stage: with task_token_format == "emb" (the default, and what every PT
    checkpoint's config carries) it splices the RAW (num_tokens, H) depth / seg parameters — 576 + 576 + 8 rows behind each image —
    while "expand_emb" mean-pools them to num_task_tokens rows like ola_arch.py.  Both run here (config.task_token_layout "raw" /
    "pooled": splice.host_plan, Engine._embed and the token-gradient scatter; golden tests/golden/tiny_llama_ift_tok.npz from the
    reference's own LlavaLlamaForCausalLM).  "text" calls embed_tokens on the float parameters (:257-258, :284-285): the reference
    itself raises there (F.embedding wants integer indices; recorded by oracle/gen_golden.py), so it is refused with that message."""

    @staticmethod
    def _check_task_token_format(cfg):
        fmt = getattr(cfg, "task_token_format", "emb")
        if cfg.num_task_tokens > 0 and cfg.token_order:
            if fmt == "text":
                raise ValueError("task_token_format='text' embeds the FLOAT special_*_tokens parameters through embed_tokens "
                                 "(llava_arch.py:257-258): the reference raises 'Expected tensor for argument #1 indices to have one of the "
                                 "following scalar types: Long, Int' in F.embedding — there is no working 'text' layout to reproduce")
            if fmt not in ("emb", "expand_emb"):
                raise ValueError(f"Unexpected task_token_format: {fmt}")                     # llava_arch.py:265
        cfg.task_token_layout = "pooled" if fmt == "expand_emb" else "raw"


class LlavaLlamaModel(LlavaMetaModel, ParamTree):
    config_class = LlavaConfig


class LlavaPhi3Model(LlavaMetaModel, ParamTree):
    config_class = LlavaPhi3Config


class _LlavaCausalLMBase(LlavaMetaForCausalLM, EngineModule):
    model_cls = LlavaLlamaModel

    def __init__(self, config, device="cuda", dtype=torch.bfloat16, init="random", seed=0):
        config.aux_heads = False
        self._check_task_token_format(config)           # sets config.task_token_layout before the parameters / engine are laid out
        EngineModule.__init__(self, config, device=device, dtype=dtype, init=init, seed=seed)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None, image_sizes=None,
                return_dict=None, **kwargs):
        """llava_llama.py:73-119 signature -> CausalLMOutputWithPast.  Outputs as language_model._run_engine: the reference's (fp32 logits
        always, hidden states on request) unless `config.reference_outputs = False` (lean mode: labelled rows only through lm_head + CE)."""
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("the MI355X path covers the training forward (input_ids + images); generation is out of scope")
        self._sync_trainable()
        eng = self._get_engine()
        if images is None:
            # text-only batch: the reference's dataset attaches a zero image to samples without one (train.py LazySupervisedDataset:
            # `data_dict['image'] = torch.zeros(3, crop, crop)` when the run is multimodal) and the splice consumes an empty feature slice
            # for samples without an <image> token (llava_arch.py:347-354).  Same here: one zero image per sample, no <image> token allowed.
            from ..config import IMAGE_TOKEN_INDEX
            if bool((input_ids == IMAGE_TOKEN_INDEX).any()):
                raise ValueError("input_ids carry an <image> token but no `images` tensor was given")
            side = self.config.cnx_image if self.config.is_convnext else self.config.vit_image
            images = torch.zeros(input_ids.shape[0], 3, side, side, device=eng.dev, dtype=torch.bfloat16)
        images, gsz = eng.image_groups(images)                       # list / 5-D `images` (llava_arch.py: same branch as ola_arch.py:262-275)
        batch = dict(input_ids=input_ids, attention_mask=attention_mask, labels=labels, images=images.to(eng.dev),
                     images_resident=bool(kwargs.get("images_resident", False)) and images.device == eng.dev)
        if gsz is not None:
            batch["image_group_sizes"] = gsz
        loss, out, logits, hidden_states = _run_engine(self, eng, batch, labels, output_hidden_states, kwargs, force_states=False)   # HF LlamaForCausalLM: states on request
        if return_dict is None:                                      # HF LlamaForCausalLM.forward: `return_dict if not None else config.use_return_dict`
            return_dict = bool(getattr(self.config, "use_return_dict", True))
        if not return_dict:                                          # llava_llama.py -> HF LlamaForCausalLM: (loss,) + (logits,) + outputs[1:]; states on request only
            want_hs = bool(output_hidden_states) or bool(getattr(self.config, "output_hidden_states", False))
            res = CausalLMOutputWithPast(loss=loss, logits=logits)
            return tuple(v for v in (loss, res.logits, hidden_states if want_hs else None) if v is not None)
        return CausalLMOutputWithPast(loss=loss, logits=logits, hidden_states=hidden_states)


class LlavaLlamaForCausalLM(_LlavaCausalLMBase):
    config_class = LlavaConfig
    model_cls = LlavaLlamaModel


class LlavaPhi3ForCausalLM(_LlavaCausalLMBase):
    config_class = LlavaPhi3Config
    model_cls = LlavaPhi3Model
