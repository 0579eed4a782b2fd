"""Configuration object for the VisPer-LM PT step.  Mirrors the attribute names the reference copies onto
`model.config` (ola_vlm/train/ola_vlm_train.py:1123-1229) plus the HF Llama / Phi-3 / CLIP hyper-parameters
(hard-coded presets: the HF hub configs are not part of the reference and there is no network)."""
from __future__ import annotations

import copy

IGNORE_INDEX = -100        # ola_vlm/constants.py:7
IMAGE_TOKEN_INDEX = -200   # ola_vlm/constants.py:8


class VisperConfig:
    """Attribute bag with HF-config-like semantics (`hasattr`, `getattr(cfg, k, default)`, `to_dict`)."""

    model_type = "ola_llama"

    def __init__(self, **kw):
        d = dict(
            arch="llama",
            vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
            num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=500000.0,
            sliding_window=None, max_position_embeddings=8192,
            # which HF release defines the sliding-window mask: False = transformers 5.x (keys with q-k < window), True = 4.41.1, the
            # reference's pin (README.md:57; q-k <= window, i.e. window+1 keys: AttentionMaskConverter tril(diagonal=-window-1))
            sliding_window_inclusive=False,
            # CLIP-ViT-L/14-336 tower (multimodal_encoder/clip_encoder.py)
            mm_vision_tower="openai/clip-vit-large-patch14-336",
            vit_hidden=1024, vit_inter=4096, vit_layers=24, vit_heads=16, vit_image=336, vit_patch=14, vit_eps=1e-5,
            mm_vision_select_layer=-2, mm_vision_select_feature="patch", mm_projector_type="mlp2x_gelu",
            mm_hidden_size=1024,
            # CLIP-ConvNeXt-XXL tower (clip_convnext_encoder.py:92-101), used when "convnext" is in mm_vision_tower
            cnx_dims=(384, 768, 1536, 3072), cnx_depths=(3, 4, 30, 3), cnx_eps=1e-5, cnx_image=768,
            # distillation (ola_vlm_train.py:1149-1229)
            aux_mode="gen-depth-seg", num_task_tokens=8, contrastive_loss_weight=0.3, use_contrastive=True,
            pass_text_to_aux=True, task_token_format="emb",
            # how append_special_tokens lays the task tokens into the sequence: "pooled" = PT stage, ola_arch.py:224-254 (depth / seg parameters
            # mean-pooled to num_task_tokens rows whatever task_token_format says); "raw" = IFT stage with task_token_format "emb",
            # llava_arch.py:259-260 (every row of the (num_tokens, H) depth / seg parameters).  Set by the model classes, not by users.
            task_token_layout="pooled",
            aux_heads=True,        # False: task tokens are spliced but no distillation heads exist (the IFT-stage LlavaLlamaForCausalLM)
            image_gen=dict(depth=1, dim_head=32, num_heads=4, num_tokens=1, output_dim=1024, ff_mult=1,
                           img_layer_indices="20", img_loss_weight=0.5),
            image_depth=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1024, ff_mult=1,
                             depth_layer_indices="18", depth_loss_weight=0.5),
            image_seg=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1536, ff_mult=1,
                           seg_layer_indices="18", seg_loss_weight=0.5),
            tokenizer_model_max_length=4096, tokenizer_padding_side="right",
            zero_masks=False,      # True reproduces the as-released `mask.zero_()` (base_ola_vlm.py:472-473,...)
            grad_reduce_dtype="bf16",   # DP gradient buckets on the wire: "bf16" (the reference's ZeRO-2 reduces bf16 gradients) | "fp32"
            train_llm=False,       # True = IFT-stage trainability: the whole LLM gets weight gradients + AdamW
            depth_decoder=False,   # True also runs the frozen DPT decoder on every depth head (base_ola_vlm.py:462-470 -> depth_preds)
        )
        d.update(kw)
        if "mm_hidden_size" not in kw:
            d["mm_hidden_size"] = d["cnx_dims"][-1] if "convnext" in str(d["mm_vision_tower"]).lower() else d["vit_hidden"]
        self.__dict__.update(d)

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def __repr__(self):
        return f"VisperConfig({self.__dict__})"

    @property
    def is_convnext(self):
        return "convnext" in str(self.mm_vision_tower).lower()

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def num_sys_tokens(self):
        """ola_llama.py:65-69 / ola_phi3.py:68."""
        if self.arch == "phi3":
            return 13
        return 26 if self.vocab_size < 128000 else 38

    @property
    def token_order(self):
        return self.aux_mode.split("-") if self.aux_mode else []


def llama3_8b(**kw) -> VisperConfig:
    """BASELINE.json configs[1]: CLIP-ViT-L/14-336 + Llama-3-8B, one distillation layer per task (d18_s18_g20)."""
    return VisperConfig(**kw)


def llama3_8b_convnext(**kw) -> VisperConfig:
    """BASELINE.json configs[3]: CLIP-ConvNeXt-XXL (768 px -> 576 x 3072) + Llama-3-8B."""
    d = dict(mm_vision_tower="laion/CLIP-convnext_xxlarge-laion2B-s34B-b82K-augreg-soup-res768")
    d.update(kw)
    return VisperConfig(**d)


def phi3_mini(**kw) -> VisperConfig:
    """BASELINE.json configs[4]: Phi-3-mini-4k (H 3072, 32 MHA heads x 96, FF 8192, V 32064, theta 1e4, window 2047)."""
    d = dict(arch="phi3", vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=32,
             num_attention_heads=32, num_key_value_heads=32, rope_theta=10000.0, sliding_window=2047,
             sliding_window_inclusive=True, max_position_embeddings=4096)
    d.update(kw)
    c = VisperConfig(**d)
    c.model_type = "ola_phi3"
    return c


def task_token_rows(cfg) -> list:
    """[(task, rows, pooled)] in token_order: how many sequence rows each task's tokens take behind an image (append_special_tokens,
    ola_arch.py:224-254 / llava_arch.py:250-293).  gen: always its num_task_tokens raw rows; depth / seg: num_task_tokens pooled rows
    (PT stage, and the IFT stage's "expand_emb") or all num_tokens raw rows (IFT stage, "emb")."""
    nt = int(cfg.num_task_tokens)
    if nt <= 0:
        return []
    raw = getattr(cfg, "task_token_layout", "pooled") == "raw"
    out = []
    for task in cfg.token_order:
        if task == "gen":
            out.append((task, nt, False))
        else:
            hc = {"depth": getattr(cfg, "image_depth", None), "seg": getattr(cfg, "image_seg", None)}[task]
            out.append((task, int(hc["num_tokens"]) if raw else nt, not raw))
    return out


def layer_indices(spec) -> list:
    """base_ola_vlm.py:97-102: '18-20' -> [17, 19]."""
    return [int(i) - 1 for i in str(spec).split("-")]
