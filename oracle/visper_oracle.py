"""CPU oracle for the VisPer-LM PT train step (TEST INFRASTRUCTURE — never shipped, never measured
as the product).

A plain-PyTorch (CPU, any float dtype) restatement of the reference's pre-training forward:
CLIP-ViT tower -> mlp2x_gelu projector -> image/task-token splice -> Llama/Phi-3 decoder ->
lm_head + NTP cross-entropy -> TaskTokenResampler heads at selected decoder layers ->
smooth-L1 + InfoNCE embedding losses.  Backward comes from torch autograd over these functions.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`visper_lm_amd`) must never import it.

Every function cites the reference file:line it restates (paths relative to /root/reference,
or the installed HF transformers copy for third-party math).

Pinning: checked against golden vectors produced by importing the reference itself in the build
container (`oracle/gen_golden.py` -> `tests/golden/*.npz`, test: `tests/test_oracle_golden.py`).
The ConvNeXt tower is NOT restated here (timm/open_clip absent: "parity unpinned").

Weights are a flat dict {reference state-dict name: tensor}.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100          # ola_vlm/constants.py:7
IMAGE_TOKEN_INDEX = -200     # ola_vlm/constants.py:8


# ----------------------------------------------------------------------------------------------
# configuration helpers
# ----------------------------------------------------------------------------------------------
def make_config(**kw) -> SimpleNamespace:
    """Config with the keys the reference copies onto `model.config`
    (ola_vlm/train/ola_vlm_train.py:1123-1229) + HF Llama/Phi-3/CLIP hyper-parameters."""
    d = dict(
        arch="llama",                     # "llama" | "phi3"
        vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
        num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=500000.0,
        sliding_window=None,
        sliding_window_inclusive=False,   # False: transformers 5.x mask (keys with q-k < window; the goldens were generated under 5.x);
                                          # True: transformers 4.41.1, the reference's pin (q-k <= window: tril diagonal = -window-1)
        # vision tower (CLIP-ViT-L/14-336)
        vit_hidden=1024, vit_inter=4096, vit_layers=24, vit_heads=16, vit_image=336, vit_patch=14,
        vit_eps=1e-5, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
        mm_projector_type="mlp2x_gelu",
        cnx_dims=(384, 768, 1536, 3072), cnx_depths=(3, 4, 30, 3), cnx_eps=1e-5, cnx_image=768,   # CLIP-ConvNeXt-XXL (config 4)
        # distillation
        aux_mode="gen-depth-seg", num_task_tokens=8, contrastive_loss_weight=0.3,
        use_contrastive=True, pass_text_to_aux=True,
        aux_heads=True,                   # False: the IFT-stage LlavaLlamaForCausalLM (task tokens may be spliced, no distillation heads)
        task_token_layout="pooled",       # "pooled": ola_arch.py:224-254 (PT) / llava_arch.py "expand_emb"; "raw": llava_arch.py:259-260 "emb" (IFT)
        image_gen=dict(depth=1, dim_head=32, num_heads=4, num_tokens=1, output_dim=1024, ff_mult=1,
                       img_layer_indices="20", img_loss_weight=0.5),
        image_depth=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1024, ff_mult=1,
                         depth_layer_indices="18", depth_loss_weight=0.5),
        image_seg=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1536, ff_mult=1,
                       seg_layer_indices="18", seg_loss_weight=0.5),
        tokenizer_model_max_length=4096, tokenizer_padding_side="right",
        zero_masks=False,                 # True = as-released `mask.zero_()` quirk
    )
    d.update(kw)
    return SimpleNamespace(**d)


def num_sys_tokens(cfg) -> int:
    """ola_llama.py:65-69 (38 Llama-3 / 26 small-vocab) ; ola_phi3.py:68 (13)."""
    if cfg.arch == "phi3":
        return 13
    return 26 if cfg.vocab_size < 128000 else 38


def layer_indices(spec: str) -> List[int]:
    """base_ola_vlm.py:97-102 — '18-20' -> [17, 19] (1-based CLI -> 0-based into layer_states)."""
    return [int(i) - 1 for i in str(spec).split("-")]


# ----------------------------------------------------------------------------------------------
# CLIP ViT tower  (clip_encoder.py:37-59 -> HF modeling_clip.py CLIPVisionTransformer)
# ----------------------------------------------------------------------------------------------
def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def vit_prefix(W) -> str:
    """transformers==4.41.1 (reference pin) nests CLIPVisionModel.vision_model; 5.x flattens it."""
    a = "model.vision_tower.vision_tower.vision_model."
    return a if (a + "embeddings.class_embedding") in W else "model.vision_tower.vision_tower."


def clip_vit_features(images: torch.Tensor, W: Dict[str, torch.Tensor], cfg, prefix=None) -> torch.Tensor:
    """Frozen CLIP-ViT forward up to hidden_states[select_layer], CLS dropped.

    clip_encoder.py:37-45 (feature_select), :47-59 (forward, no_grad);
    HF CLIPVisionEmbeddings (patch conv stride=patch, no bias; CLS; learned pos-emb),
    pre_layrnorm, CLIPEncoderLayer (LN -> MHA(+bias, scale d^-0.5) -> +res -> LN -> fc1 ->
    quick_gelu -> fc2 -> +res).  hidden_states[k] = input of layer k, so select_layer=-2 needs
    the first (L-1) layers only."""
    p = prefix or vit_prefix(W)
    dt = W[p + "embeddings.patch_embedding.weight"].dtype
    x = images.to(dt)
    B = x.shape[0]
    pe = F.conv2d(x, W[p + "embeddings.patch_embedding.weight"], stride=cfg.vit_patch)   # (B,C,g,g)
    pe = pe.flatten(2).transpose(1, 2)
    cls = W[p + "embeddings.class_embedding"].expand(B, 1, -1)
    h = torch.cat([cls, pe], dim=1) + W[p + "embeddings.position_embedding.weight"][None]
    C = h.shape[-1]
    h = F.layer_norm(h, (C,), W[p + "pre_layrnorm.weight"], W[p + "pre_layrnorm.bias"], cfg.vit_eps)
    nh = cfg.vit_heads
    hd = C // nh
    sel = cfg.mm_vision_select_layer
    n_run = cfg.vit_layers + 1 + sel if sel < 0 else sel      # index into hidden_states
    for l in range(n_run):
        q = p + f"encoder.layers.{l}."
        r = h
        y = F.layer_norm(h, (C,), W[q + "layer_norm1.weight"], W[q + "layer_norm1.bias"], cfg.vit_eps)
        N = y.shape[1]
        qq = F.linear(y, W[q + "self_attn.q_proj.weight"], W[q + "self_attn.q_proj.bias"]).view(B, N, nh, hd).transpose(1, 2)
        kk = F.linear(y, W[q + "self_attn.k_proj.weight"], W[q + "self_attn.k_proj.bias"]).view(B, N, nh, hd).transpose(1, 2)
        vv = F.linear(y, W[q + "self_attn.v_proj.weight"], W[q + "self_attn.v_proj.bias"]).view(B, N, nh, hd).transpose(1, 2)
        att = torch.matmul(qq, kk.transpose(-1, -2)) * (hd ** -0.5)
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(qq.dtype)
        o = torch.matmul(att, vv).transpose(1, 2).reshape(B, N, C)
        o = F.linear(o, W[q + "self_attn.out_proj.weight"], W[q + "self_attn.out_proj.bias"])
        h = r + o
        r = h
        y = F.layer_norm(h, (C,), W[q + "layer_norm2.weight"], W[q + "layer_norm2.bias"], cfg.vit_eps)
        y = quick_gelu(F.linear(y, W[q + "mlp.fc1.weight"], W[q + "mlp.fc1.bias"]))
        y = F.linear(y, W[q + "mlp.fc2.weight"], W[q + "mlp.fc2.bias"])
        h = r + y
    if cfg.mm_vision_select_feature == "patch":
        h = h[:, 1:]
    return h.to(images.dtype)


def convnext_features(images, W, cfg, prefix="model.vision_tower.vision_tower."):
    """CLIP-ConvNeXt trunk exactly as the reference drives it (clip_convnext_encoder.py:150-174): stem -> stages ->
    norm_pre (identity for the CLIP trunks) -> flatten(2,3).permute(0,2,1).
    UNPINNED: timm / open_clip are not installed and there is no network, so this follows the public ConvNeXt definition
    (timm convnext.py: stem = Conv 4x4/s4 + LayerNorm2d; stage i>0 starts with LayerNorm2d + Conv 2x2/s2; block =
    dwconv7x7 -> LayerNorm -> Linear(C,4C) -> GELU -> Linear(4C,C) -> * gamma -> + shortcut; norm_eps 1e-5 for xxlarge)."""
    p = prefix
    eps = cfg.cnx_eps

    def ln2d(x, w, b):
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)
    x = F.conv2d(images.to(W[p + "stem.0.weight"].dtype), W[p + "stem.0.weight"], W[p + "stem.0.bias"], stride=4)
    x = ln2d(x, W[p + "stem.1.weight"], W[p + "stem.1.bias"])
    for i, depth in enumerate(cfg.cnx_depths):
        q = f"{p}stages.{i}."
        if i > 0:
            x = ln2d(x, W[q + "downsample.0.weight"], W[q + "downsample.0.bias"])
            x = F.conv2d(x, W[q + "downsample.1.weight"], W[q + "downsample.1.bias"], stride=2)
        C = x.shape[1]
        for j in range(depth):
            b = f"{q}blocks.{j}."
            y = F.conv2d(x, W[b + "conv_dw.weight"], W[b + "conv_dw.bias"], padding=3, groups=C)
            y = y.permute(0, 2, 3, 1)
            y = F.layer_norm(y, (C,), W[b + "norm.weight"], W[b + "norm.bias"], eps)
            y = F.linear(F.gelu(F.linear(y, W[b + "mlp.fc1.weight"], W[b + "mlp.fc1.bias"])), W[b + "mlp.fc2.weight"], W[b + "mlp.fc2.bias"])
            y = y * W[b + "gamma"]
            x = x + y.permute(0, 3, 1, 2)
    return x.flatten(2, 3).permute(0, 2, 1).contiguous().to(images.dtype)


def mm_projector(x, W, prefix="model.mm_projector."):
    """multimodal_projector/builder.py:53-60 mlp2x_gelu: Linear -> GELU(erf) -> Linear."""
    y = F.linear(x, W[prefix + "0.weight"], W[prefix + "0.bias"])
    y = F.gelu(y)
    return F.linear(y, W[prefix + "2.weight"], W[prefix + "2.bias"])


def encode_images(images, W, cfg):
    """ola_arch.py:187-190."""
    with torch.no_grad():
        if "model.vision_tower.vision_tower.stem.0.weight" in W:
            feats = convnext_features(images, W, cfg)
        else:
            feats = clip_vit_features(images, W, cfg)
    return mm_projector(feats.to(images.dtype), W)


# ----------------------------------------------------------------------------------------------
# sequence splice  (ola_arch.py:224-254, 256-444)
# ----------------------------------------------------------------------------------------------
def task_token_rows(W, cfg) -> List[torch.Tensor]:
    """append_special_tokens.  PT stage (ola_arch.py:224-254): per task in token_order, 8 rows: depth/seg = mean over (num_tokens/8)-row
    groups of the (num_tokens,H) parameter; gen = raw.  IFT stage (llava_arch.py:250-293): task_token_format "expand_emb" = the same
    pooling, "emb" (the default, :259-260) = every row of the depth / seg parameters as it is (layout "raw"); "text" calls embed_tokens on
    the FLOAT parameters (:257-258, :284-285), which raises in F.embedding — the reference has no working "text" path."""
    rows = []
    n = cfg.num_task_tokens
    raw = getattr(cfg, "task_token_layout", "pooled") == "raw"
    for t in (cfg.aux_mode.split("-") if cfg.aux_mode else []):
        name = f"model.special_{t}_tokens"
        if n > 0 and name in W:
            tk = W[name]
            if t in ("depth", "seg") and not raw:
                tk = tk.view(n, tk.shape[0] // n, tk.shape[1]).mean(dim=1)
            rows.append(tk)
    return rows


def prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, image_features, W, cfg):
    """ola_arch.py:256-444 restated; `image_features` is indexed per <image> token / text-only sample: a [n, 576, H] tensor (stacked 4-D
    `images`) or the list of flattened [n_j * 576, H] groups of the list / 5-D "flat" merge (:262-275)).

    Returns (position_ids, attention_mask, inputs_embeds, labels) after splice / truncate / pad."""
    embed = W["model.embed_tokens.weight"]
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    attention_mask = attention_mask.bool()
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    new_embeds, new_labels = [], []
    img_idx = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b]]
        lab = labels[b][attention_mask[b]]
        pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
        if len(pos) == 0:
            new_embeds.append(torch.cat([embed[ids], image_features[img_idx][0:0]], dim=0))
            new_labels.append(lab)
            img_idx += 1
            continue
        bounds = [-1] + pos + [ids.shape[0]]
        pe, pl = [], []
        for i in range(len(bounds) - 1):
            seg_ids = ids[bounds[i] + 1: bounds[i + 1]]
            pe.append(embed[seg_ids])
            pl.append(lab[bounds[i] + 1: bounds[i + 1]])
            if i < len(pos):
                f = image_features[img_idx]
                img_idx += 1
                pe.append(f)
                pl.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=lab.dtype))
                for tk in task_token_rows(W, cfg):
                    pe.append(tk.to(f.dtype))
                    pl.append(torch.full((tk.shape[0],), IGNORE_INDEX, dtype=lab.dtype))
        new_embeds.append(torch.cat(pe, dim=0))
        new_labels.append(torch.cat(pl, dim=0))
    mx = cfg.tokenizer_model_max_length
    if mx is not None:
        new_embeds = [x[:mx] for x in new_embeds]
        new_labels = [x[:mx] for x in new_labels]
    L = max(x.shape[0] for x in new_embeds)
    H = new_embeds[0].shape[1]
    out = torch.zeros(B, L, H, dtype=new_embeds[0].dtype)
    out_l = torch.full((B, L), IGNORE_INDEX, dtype=labels.dtype)
    am = torch.zeros(B, L, dtype=torch.bool)
    pid = torch.zeros(B, L, dtype=torch.long)
    left = cfg.tokenizer_padding_side == "left"
    rows = []
    for b, (e, l) in enumerate(zip(new_embeds, new_labels)):
        n = e.shape[0]
        padz = torch.zeros(L - n, H, dtype=e.dtype)
        if left:
            rows.append(torch.cat([padz, e], 0))
            if n > 0:
                out_l[b, L - n:] = l; am[b, L - n:] = True; pid[b, L - n:] = torch.arange(n)
        else:
            rows.append(torch.cat([e, padz], 0))
            if n > 0:
                out_l[b, :n] = l; am[b, :n] = True; pid[b, :n] = torch.arange(n)
    return pid, am, torch.stack(rows, 0), out_l


# ----------------------------------------------------------------------------------------------
# decoder  (ola_llama.py:105-119 -> HF LlamaModel / Phi3Model)
# ----------------------------------------------------------------------------------------------
def rms_norm(x, w, eps):
    """HF LlamaRMSNorm (modeling_llama.py:53-68): fp32 variance, cast back, * weight."""
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_tables(position_ids, head_dim, theta, dtype):
    """HF LlamaRotaryEmbedding.forward: fp32 angles, cos/sin cast to the activation dtype."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def decoder_forward(inputs_embeds, position_ids, attention_mask, W, cfg, prefix="model."):
    """32x [RMSNorm -> QKV -> RoPE -> causal softmax(fp32) attention -> O -> +res -> RMSNorm ->
    SwiGLU -> +res], final RMSNorm.  Returns (hidden_post_norm, layer_states) with
    layer_states[i] = residual stream after layer i+1, the LAST one replaced by the post-norm
    state — exactly `outputs[-1][1:]` of HF with output_hidden_states=True (ola_llama.py:117-119)."""
    x = inputs_embeds
    B, S, H = x.shape
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    hd = H // nh
    if position_ids is None:
        position_ids = torch.arange(S)[None].expand(B, S)
    cos, sin = rope_tables(position_ids, hd, cfg.rope_theta, x.dtype)
    cos, sin = cos[:, None], sin[:, None]
    neg = torch.finfo(x.dtype).min
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    if cfg.sliding_window is not None:
        # HF 4.41.1 AttentionMaskConverter._make_causal_mask masks tril(diagonal = -window - 1) -> window + 1 visible keys;
        # HF 5.x masking_utils.sliding_window_overlay keeps kv > q - window -> window visible keys
        causal = causal & ~torch.ones(S, S, dtype=torch.bool).tril(-(int(cfg.sliding_window) + int(getattr(cfg, "sliding_window_inclusive", False))))
    allow = causal[None, None]
    if attention_mask is not None:
        allow = allow & attention_mask.bool()[:, None, None, :]
    bias = torch.zeros(allow.shape, dtype=x.dtype).masked_fill(~allow, neg)
    states = []
    for l in range(cfg.num_hidden_layers):
        p = f"{prefix}layers.{l}."
        r = x
        y = rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        if cfg.arch == "phi3":
            qkv = F.linear(y, W[p + "self_attn.qkv_proj.weight"])
            q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
        else:
            q = F.linear(y, W[p + "self_attn.q_proj.weight"])
            k = F.linear(y, W[p + "self_attn.k_proj.weight"])
            v = F.linear(y, W[p + "self_attn.v_proj.weight"])
        q = q.view(B, S, nh, hd).transpose(1, 2)
        k = k.view(B, S, nkv, hd).transpose(1, 2)
        v = v.view(B, S, nkv, hd).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        rep = nh // nkv
        if rep > 1:
            k = k[:, :, None].expand(B, nkv, rep, S, hd).reshape(B, nh, S, hd)
            v = v[:, :, None].expand(B, nkv, rep, S, hd).reshape(B, nh, S, hd)
        att = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5) + bias
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(att, v).transpose(1, 2).reshape(B, S, nh * hd)
        x = r + F.linear(o, W[p + "self_attn.o_proj.weight"])
        r = x
        y = rms_norm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        if cfg.arch == "phi3":
            gu = F.linear(y, W[p + "mlp.gate_up_proj.weight"])
            g, u = gu.chunk(2, dim=-1)
        else:
            g = F.linear(y, W[p + "mlp.gate_proj.weight"])
            u = F.linear(y, W[p + "mlp.up_proj.weight"])
        x = r + F.linear(F.silu(g) * u, W[p + "mlp.down_proj.weight"])
        states.append(x)
    hidden = rms_norm(x, W[prefix + "norm.weight"], cfg.rms_norm_eps)
    states[-1] = hidden
    return hidden, states


def ntp_loss(hidden, labels, W, cfg, keep_logits=True, chunk_rows=2048):
    """ola_llama.py:121-136: lm_head (no bias) -> .float() -> shifted CE, mean over labels != -100.
    keep_logits=False: the same arithmetic in row chunks (every row still goes through lm_head like in the reference; only the
    8.4 GB fp32 logits tensor of configs[1] is not held at once): sum of the per-row losses / number of labelled rows."""
    if keep_logits:
        logits = F.linear(hidden, W["lm_head.weight"]).float()
        sl = logits[:, :-1].reshape(-1, logits.shape[-1])
        tl = labels[:, 1:].reshape(-1)
        return logits, F.cross_entropy(sl, tl, ignore_index=IGNORE_INDEX)
    h = hidden[:, :-1].reshape(-1, hidden.shape[-1])
    tl = labels[:, 1:].reshape(-1)
    tot = hidden.new_zeros((), dtype=torch.float32)
    for r0 in range(0, h.shape[0], chunk_rows):
        lg = F.linear(h[r0:r0 + chunk_rows], W["lm_head.weight"]).float()
        tot = tot + F.cross_entropy(lg, tl[r0:r0 + chunk_rows], ignore_index=IGNORE_INDEX, reduction="sum")
    return None, tot / (tl != IGNORE_INDEX).sum()


# ----------------------------------------------------------------------------------------------
# heads  (resampler.py:9-75,167-224; gen_head.py:39-65; oneformer_head.py:224-258; da_v2_head.py:418-457)
# ----------------------------------------------------------------------------------------------
def task_token_resampler(x, latents, W, pfx, hcfg):
    """TaskTokenResampler.forward (resampler.py:202-224) with depth==len(layers) Perceiver blocks."""
    nq = hcfg["num_tokens"]
    heads = hcfg["num_heads"]
    dh = hcfg["dim_head"]
    if latents.shape[1] != nq:
        if nq > 1 and nq % latents.shape[1] == 0:
            latents = latents.repeat(1, nq // latents.shape[1], 1)
        else:
            latents = latents.mean(dim=1, keepdim=True).repeat(1, nq, 1)
    lat = F.linear(latents, W[pfx + "proj_in.weight"], W[pfx + "proj_in.bias"])
    x = F.linear(x, W[pfx + "proj_in.weight"], W[pfx + "proj_in.bias"])
    D = lat.shape[-1]
    for d in range(hcfg["depth"]):
        a = f"{pfx}layers.{d}.0."
        f = f"{pfx}layers.{d}.1."
        # PerceiverAttention.forward resampler.py:46-75
        xn = F.layer_norm(x, (D,), W[a + "norm1.weight"], W[a + "norm1.bias"])
        ln = F.layer_norm(lat, (D,), W[a + "norm2.weight"], W[a + "norm2.bias"])
        b, l, _ = ln.shape
        q = F.linear(ln, W[a + "to_q.weight"])
        kv = F.linear(torch.cat([xn, ln], dim=-2), W[a + "to_kv.weight"])
        k, v = kv.chunk(2, dim=-1)
        sp = lambda t: t.view(b, t.shape[1], heads, -1).transpose(1, 2)
        q, k, v = sp(q), sp(k), sp(v)
        sc = 1.0 / math.sqrt(math.sqrt(dh))
        w = (q * sc) @ (k * sc).transpose(-2, -1)
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)
        o = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
        lat = F.linear(o, W[a + "to_out.weight"]) + lat
        # FeedForward resampler.py:9-16: LN -> Linear(no bias) -> GELU -> Linear(no bias)
        y = F.layer_norm(lat, (D,), W[f + "0.weight"], W[f + "0.bias"])
        y = F.linear(F.gelu(F.linear(y, W[f + "1.weight"])), W[f + "3.weight"])
        lat = y + lat
    out = F.linear(lat, W[pfx + "proj_out.weight"], W[pfx + "proj_out.bias"])
    Do = out.shape[-1]
    return F.layer_norm(out, (Do,), W[pfx + "norm_out.weight"], W[pfx + "norm_out.bias"])


def resampler(x, W, pfx, hcfg):
    """Resampler.forward (resampler.py:150-165), the num_task_tokens == 0 heads' projector: the queries are the module's own `latents`
    parameter (1, num_queries, dim), tiled over the batch and NOT passed through proj_in; x is."""
    heads, dh = hcfg["num_heads"], hcfg["dim_head"]
    lat = W[pfx + "latents"].repeat(x.shape[0], 1, 1).to(x.dtype)
    x = F.linear(x, W[pfx + "proj_in.weight"], W[pfx + "proj_in.bias"])
    D = lat.shape[-1]
    for d in range(hcfg["depth"]):
        a, f = f"{pfx}layers.{d}.0.", f"{pfx}layers.{d}.1."
        xn = F.layer_norm(x, (D,), W[a + "norm1.weight"], W[a + "norm1.bias"])                  # PerceiverAttention.forward :46-75
        ln = F.layer_norm(lat, (D,), W[a + "norm2.weight"], W[a + "norm2.bias"])
        b, l, _ = ln.shape
        q = F.linear(ln, W[a + "to_q.weight"])
        k, v = F.linear(torch.cat([xn, ln], dim=-2), W[a + "to_kv.weight"]).chunk(2, dim=-1)
        sp = lambda t: t.view(b, t.shape[1], heads, -1).transpose(1, 2)
        q, k, v = sp(q), sp(k), sp(v)
        sc = 1.0 / math.sqrt(math.sqrt(dh))
        w = (q * sc) @ (k * sc).transpose(-2, -1)
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)
        lat = F.linear((w @ v).permute(0, 2, 1, 3).reshape(b, l, -1), W[a + "to_out.weight"]) + lat
        y = F.layer_norm(lat, (D,), W[f + "0.weight"], W[f + "0.bias"])                          # FeedForward :9-16
        lat = F.linear(F.gelu(F.linear(y, W[f + "1.weight"])), W[f + "3.weight"]) + lat
    out = F.linear(lat, W[pfx + "proj_out.weight"], W[pfx + "proj_out.bias"])
    return F.layer_norm(out, (out.shape[-1],), W[pfx + "norm_out.weight"], W[pfx + "norm_out.bias"])


def head_inputs(state, task, W, cfg):
    """forward_emb_predictor token selection (base_ola_vlm.py:413-441); latents None when num_task_tokens == 0 (:429-430)."""
    order = cfg.aux_mode.split("-")
    ns = num_sys_tokens(cfg)
    nt = cfg.num_task_tokens
    k = order.index(task)
    s0 = ns + 576 + nt * k
    end = ns + 576 + nt * len(order)
    x = state[:, :ns + 576]
    if nt == 0 or state.shape[1] < 600:
        if cfg.pass_text_to_aux:
            x = state
    else:
        x = torch.cat([x, state[:, s0:s0 + nt]], dim=1)
        if cfg.pass_text_to_aux:
            x = torch.cat([x, state[:, end:]], dim=1)
    if nt == 0:
        return x, None
    if task != "gen":
        lat = W[f"model.special_{task}_tokens"][None].repeat(x.shape[0], 1, 1)
    else:
        lat = x[:, -nt:] if not cfg.pass_text_to_aux else x[:, ns + 576: ns + 576 + nt]
    return x, lat


def _mlp_relu(x, W, p):
    """da_v2_head.py:331-335 build_mlp: Linear -> ReLU -> Linear."""
    return F.linear(F.relu(F.linear(x, W[p + "0.weight"], W[p + "0.bias"])), W[p + "2.weight"], W[p + "2.bias"])


def head_forward(state, task, i, W, cfg):
    """One head instance i of `task` on one layer state -> (loss_input_pred, extras)."""
    x, lat = head_inputs(state, task, W, cfg)
    name = {"gen": "image_gen_heads", "seg": "image_seg_heads", "depth": "image_depth_heads"}[task]
    hcfg = {"gen": cfg.image_gen, "seg": cfg.image_seg, "depth": cfg.image_depth}[task]
    if lat is None:                                                   # GenHead / DepthHead / OneFormerSegHead (num_task_tokens == 0)
        v = resampler(x, W, f"{name}.{i}.projector.", hcfg)
    else:
        v = task_token_resampler(x, lat.to(x.dtype), W, f"{name}.{i}.projector.", hcfg)
    if task == "gen":
        return v, None
    if task == "seg":
        b, n, c = v.shape
        g = int(math.sqrt(n))
        return v.permute(0, 2, 1).reshape(b, c, g, g), None          # oneformer_head.py:250-258
    if not hcfg.get("use_intermediate_depth", True):                  # da_v2_head.py:448-455: features = [(visual_feats, None)]
        return v, [v]                                                 # loss on visual_feats itself (all_depth_feats[0][0], base_ola_vlm.py:369)
    feats = [_mlp_relu(v, W, f"{name}.{i}.linear_{j}.") for j in (1, 2, 3)] + [v]   # da_v2_head.py:444-457
    return feats[0], feats                                            # loss uses lin1(v): base_ola_vlm.py:369


# ----------------------------------------------------------------------------------------------
# a11: frozen DPT depth decoder  (aux_heads/da_v2_head.py:182-321; called under no_grad at
# base_ola_vlm.py:462-470 on the depth head's 4 feature maps; output only -> `depth_preds`)
# ----------------------------------------------------------------------------------------------
DPT_OUT_CHANNELS = (256, 512, 1024, 1024)             # da_v2_head.py:300 (vitl)
DPT_FEATURES = 256


def _rcu(x, W, p):
    """ResidualConvUnit.forward (da_v2_head.py:63-88), bn=False: conv2(relu(conv1(relu(x)))) + x."""
    out = F.conv2d(F.relu(x), W[p + "conv1.weight"], W[p + "conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), W[p + "conv2.weight"], W[p + "conv2.bias"], padding=1)
    return out + x


def _fusion(W, p, x0, x1=None, size=None):
    """FeatureFusionBlock.forward (da_v2_head.py:127-153): align_corners=True bilinear, then the 1x1 out_conv."""
    out = x0
    if x1 is not None:
        out = out + _rcu(x1, W, p + "resConfUnit1.")
    out = _rcu(out, W, p + "resConfUnit2.")
    if size is None:
        out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        out = F.interpolate(out, size=tuple(size), mode="bilinear", align_corners=True)
    return F.conv2d(out, W[p + "out_conv.weight"], W[p + "out_conv.bias"])


def dpt_depth_pred(feats, W, prefix="da_v2_head.depth_head.", patch=24):
    """DAv2_Head.forward (da_v2_head.py:316-321) + DPTHead.forward (:260-293) + the min-max normalisation of
    base_ola_vlm.py:466-468.  feats: 4 tensors (B, patch*patch, 1024) -> (B, 14*patch, 14*patch)."""
    outs = []
    for i, x in enumerate(feats):
        x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], patch, patch)
        x = F.conv2d(x, W[prefix + f"projects.{i}.weight"], W[prefix + f"projects.{i}.bias"])
        if i == 0:
            x = F.conv_transpose2d(x, W[prefix + "resize_layers.0.weight"], W[prefix + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, W[prefix + "resize_layers.1.weight"], W[prefix + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = F.conv2d(x, W[prefix + "resize_layers.3.weight"], W[prefix + "resize_layers.3.bias"], stride=2, padding=1)
        outs.append(x)
    sc = prefix + "scratch."
    rn = [F.conv2d(outs[i], W[sc + f"layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    path4 = _fusion(W, sc + "refinenet4.", rn[3], size=rn[2].shape[2:])
    path3 = _fusion(W, sc + "refinenet3.", path4, rn[2], size=rn[1].shape[2:])
    path2 = _fusion(W, sc + "refinenet2.", path3, rn[1], size=rn[0].shape[2:])
    path1 = _fusion(W, sc + "refinenet1.", path2, rn[0])
    out = F.conv2d(path1, W[sc + "output_conv1.weight"], W[sc + "output_conv1.bias"], padding=1)
    out = F.interpolate(out, (14 * patch, 14 * patch), mode="bilinear", align_corners=True)
    out = F.relu(F.conv2d(out, W[sc + "output_conv2.0.weight"], W[sc + "output_conv2.0.bias"], padding=1))
    out = F.relu(F.conv2d(out, W[sc + "output_conv2.2.weight"], W[sc + "output_conv2.2.bias"]))
    depth = F.relu(out).squeeze(1)
    mn, mx = depth.amin(dim=(1, 2), keepdim=True), depth.amax(dim=(1, 2), keepdim=True)
    return (depth - mn) / (mx - mn)


def dpt_param_shapes(prefix="da_v2_head.depth_head."):
    """State-dict names/shapes of DAv2_Head('vitl') (da_v2_head.py:182-258, 296-314)."""
    oc, f = DPT_OUT_CHANNELS, DPT_FEATURES
    sh = {}
    for i, c in enumerate(oc):
        sh[prefix + f"projects.{i}.weight"] = (c, 1024, 1, 1)
        sh[prefix + f"projects.{i}.bias"] = (c,)
    sh[prefix + "resize_layers.0.weight"] = (oc[0], oc[0], 4, 4)
    sh[prefix + "resize_layers.0.bias"] = (oc[0],)
    sh[prefix + "resize_layers.1.weight"] = (oc[1], oc[1], 2, 2)
    sh[prefix + "resize_layers.1.bias"] = (oc[1],)
    sh[prefix + "resize_layers.3.weight"] = (oc[3], oc[3], 3, 3)
    sh[prefix + "resize_layers.3.bias"] = (oc[3],)
    sc = prefix + "scratch."
    for i, c in enumerate(oc):
        sh[sc + f"layer{i + 1}_rn.weight"] = (f, c, 3, 3)
    for r in (1, 2, 3, 4):
        p = sc + f"refinenet{r}."
        sh[p + "out_conv.weight"] = (f, f, 1, 1)
        sh[p + "out_conv.bias"] = (f,)
        for u in (1, 2):
            for cv in (1, 2):
                sh[p + f"resConfUnit{u}.conv{cv}.weight"] = (f, f, 3, 3)
                sh[p + f"resConfUnit{u}.conv{cv}.bias"] = (f,)
    sh[sc + "output_conv1.weight"] = (f // 2, f, 3, 3)
    sh[sc + "output_conv1.bias"] = (f // 2,)
    sh[sc + "output_conv2.0.weight"] = (32, f // 2, 3, 3)
    sh[sc + "output_conv2.0.bias"] = (32,)
    sh[sc + "output_conv2.2.weight"] = (1, 32, 1, 1)
    sh[sc + "output_conv2.2.bias"] = (1,)
    return sh


# ----------------------------------------------------------------------------------------------
# f-3: frozen depth teacher features (base_ola_vlm.py:347-365 _get_dav2_feats -> DepthAnythingV2.forward, dpt.py:164-169 ->
# DinoVisionTransformer.get_intermediate_layers, depth_anything_v2/dinov2.py:177-330): mean of the final-normed patch tokens of
# 4 intermediate blocks of a DINOv2 ViT (LayerScale blocks, GELU(erf) MLP, LayerNorm eps 1e-6, bicubic position interpolation).
# ----------------------------------------------------------------------------------------------
def dinov2_pos_embed(pos_embed, grid, patch_offset=0.1):
    """interpolate_pos_encoding (dinov2.py:177-204): (1, 1 + N, C) -> (1, 1 + grid*grid, C); identity when the grids match."""
    N = pos_embed.shape[1] - 1
    if N == grid * grid:
        return pos_embed
    C = pos_embed.shape[-1]
    sq = int(math.sqrt(N))
    sc = float(grid + patch_offset) / math.sqrt(N)
    pp = F.interpolate(pos_embed[:, 1:].float().reshape(1, sq, sq, C).permute(0, 3, 1, 2), scale_factor=(sc, sc), mode="bicubic",
                       antialias=False)
    assert pp.shape[-1] == grid and pp.shape[-2] == grid
    return torch.cat([pos_embed[:, :1].float(), pp.permute(0, 2, 3, 1).reshape(1, -1, C)], 1).to(pos_embed.dtype)


def dinov2_depth_target(images, W, heads, taps, prefix="dav2_backbone.pretrained.", patch=14):
    """images (B,3,S,S) normalised -> (B, (S/14)^2, C): (f0 + f1 + f2 + f3) / 4 of the normed patch tokens (base_ola_vlm.py:355)."""
    B, _, S, _ = images.shape
    grid = S // patch
    x = F.conv2d(images, W[prefix + "patch_embed.proj.weight"], W[prefix + "patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    x = torch.cat([W[prefix + "cls_token"].expand(B, -1, -1).to(x.dtype), x], 1)
    x = x + dinov2_pos_embed(W[prefix + "pos_embed"], grid).to(x.dtype)
    C = x.shape[-1]
    hd = C // heads
    feats = []
    for i in range(max(taps) + 1):
        b = f"{prefix}blocks.{i}."
        y = F.layer_norm(x, (C,), W[b + "norm1.weight"], W[b + "norm1.bias"], 1e-6)
        qkv = F.linear(y, W[b + "attn.qkv.weight"], W[b + "attn.qkv.bias"]).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1), -1) @ qkv[2]
        y = F.linear(att.transpose(1, 2).reshape(B, -1, C), W[b + "attn.proj.weight"], W[b + "attn.proj.bias"])
        x = x + y * W[b + "ls1.gamma"]
        y = F.layer_norm(x, (C,), W[b + "norm2.weight"], W[b + "norm2.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, W[b + "mlp.fc1.weight"], W[b + "mlp.fc1.bias"])), W[b + "mlp.fc2.weight"], W[b + "mlp.fc2.bias"])
        x = x + y * W[b + "ls2.gamma"]
        if i in taps:
            feats.append(F.layer_norm(x, (C,), W[prefix + "norm.weight"], W[prefix + "norm.bias"], 1e-6)[:, 1:])
    return sum(feats) / len(feats)


def clip_image_embeds(images, W, heads, patch, act="gelu", eps=1e-5, prefix="pipe.image_encoder."):
    """f-3: the generation teacher target (base_ola_vlm.py:323-332: `pipe.image_encoder(x).image_embeds`, unCLIP's
    CLIPVisionModelWithProjection = CLIP ViT-H/14): full tower, `post_layernorm` on the CLS token, bias-free `visual_projection`
    (HF modeling_clip.py CLIPVisionTransformer.forward + CLIPVisionModelWithProjection.forward).  -> (B, 1, proj_dim)."""
    p = prefix + "vision_model."
    B = images.shape[0]
    pe = F.conv2d(images, W[p + "embeddings.patch_embedding.weight"], stride=patch).flatten(2).transpose(1, 2)
    h = torch.cat([W[p + "embeddings.class_embedding"].expand(B, 1, -1), pe], 1) + W[p + "embeddings.position_embedding.weight"][None]
    C = h.shape[-1]
    hd = C // heads
    h = F.layer_norm(h, (C,), W[p + "pre_layrnorm.weight"], W[p + "pre_layrnorm.bias"], eps)
    L = 1 + max(int(k.split("encoder.layers.")[1].split(".")[0]) for k in W if k.startswith(p + "encoder.layers."))
    fn = quick_gelu if act == "quick_gelu" else F.gelu
    for l in range(L):
        q = p + f"encoder.layers.{l}."
        y = F.layer_norm(h, (C,), W[q + "layer_norm1.weight"], W[q + "layer_norm1.bias"], eps)
        N = y.shape[1]
        qq, kk, vv = (F.linear(y, W[q + f"self_attn.{n}_proj.weight"], W[q + f"self_attn.{n}_proj.bias"]).view(B, N, heads, hd).transpose(1, 2)
                      for n in "qkv")
        att = torch.softmax(qq @ kk.transpose(-1, -2) * hd ** -0.5, -1)
        h = h + F.linear((att @ vv).transpose(1, 2).reshape(B, N, C), W[q + "self_attn.out_proj.weight"], W[q + "self_attn.out_proj.bias"])
        y = F.layer_norm(h, (C,), W[q + "layer_norm2.weight"], W[q + "layer_norm2.bias"], eps)
        h = h + F.linear(fn(F.linear(y, W[q + "mlp.fc1.weight"], W[q + "mlp.fc1.bias"])), W[q + "mlp.fc2.weight"], W[q + "mlp.fc2.bias"])
    pooled = F.layer_norm(h[:, 0], (C,), W[p + "post_layernorm.weight"], W[p + "post_layernorm.bias"], eps)
    return F.linear(pooled, W[prefix + "visual_projection.weight"]).unsqueeze(1)


def _swin_rel_index(ws):
    """SwinRelativePositionBias._create_relative_position_index (HF modeling_swin.py): flat index into the (2ws-1)^2 bias table."""
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _swin_shift_mask(Hh, Ww, ws, shift):
    """SwinLayer.get_attn_mask: (nW, ws*ws, ws*ws) with -100 between tokens of different cyclic-shift regions."""
    hr = (torch.arange(Hh) >= Hh - ws).long() + (torch.arange(Hh) >= Hh - shift).long()
    wr = (torch.arange(Ww) >= Ww - ws).long() + (torch.arange(Ww) >= Ww - shift).long()
    img = (hr[:, None] * 3 + wr[None, :]).float()
    mw = img.view(Hh // ws, ws, Ww // ws, ws).transpose(1, 2).reshape(-1, ws * ws)
    d = mw[:, None, :] - mw[:, :, None]
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def swin_seg_target(images, W, depths, heads, window=12, patch=4, out_hw=24, prefix="oneformer.model.pixel_level_module.encoder.", eps=1e-5):
    """f-3: the segmentation teacher target (base_ola_vlm.py:382-397 -> oneformer_head.py:11-69): Swin backbone feature_maps[-1]
    (last stage, `hidden_states_norms.stage4`), bilinear (align_corners=False) to 24 x 24.  HF modeling_swin.py: SwinEmbeddings,
    SwinLayer (W-MSA / SW-MSA with relative position bias + cyclic-shift mask, always_partition=True as SwinBackbone.forward sets),
    SwinPatchMerging.  images (B,3,S,S) -> (B, C_last, 24, 24)."""
    p = prefix + "swin."
    x = F.conv2d(images, W[p + "embeddings.patch_embeddings.projection.weight"], W[p + "embeddings.patch_embeddings.projection.bias"], stride=patch)
    B, C, Hh, Ww = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (C,), W[p + "embeddings.norm.weight"], W[p + "embeddings.norm.bias"], eps)
    ridx = _swin_rel_index(window).view(-1)
    N = window * window
    for s, (dep, nh) in enumerate(zip(depths, heads)):
        hd = C // nh
        for bi in range(dep):
            q = f"{p}encoder.layers.{s}.blocks.{bi}."
            shift = 0 if bi % 2 == 0 else window // 2
            y = F.layer_norm(x, (C,), W[q + "layernorm_before.weight"], W[q + "layernorm_before.bias"], eps).view(B, Hh, Ww, C)
            if shift:
                y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
            win = y.view(B, Hh // window, window, Ww // window, window, C).transpose(2, 3).reshape(-1, N, C)
            qq, kk, vv = (F.linear(win, W[q + f"attention.{n}_proj.weight"], W[q + f"attention.{n}_proj.bias"]).view(-1, N, nh, hd).transpose(1, 2)
                          for n in "qkv")
            bias = W[q + "attention.relative_position_bias.relative_position_bias_table"][ridx].view(N, N, nh).permute(2, 0, 1)[None]
            sc = qq @ kk.transpose(-1, -2) * hd ** -0.5 + bias
            if shift:
                sc = sc + _swin_shift_mask(Hh, Ww, window, shift).repeat(B, 1, 1)[:, None].to(sc.dtype)
            o = (torch.softmax(sc, -1) @ vv).transpose(1, 2).reshape(-1, N, C)
            o = F.linear(o, W[q + "attention.o_proj.weight"], W[q + "attention.o_proj.bias"])
            o = o.view(B, Hh // window, Ww // window, window, window, C).transpose(2, 3).reshape(B, Hh, Ww, C)
            if shift:
                o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
            x = x + o.reshape(B, Hh * Ww, C)
            y = F.layer_norm(x, (C,), W[q + "layernorm_after.weight"], W[q + "layernorm_after.bias"], eps)
            x = x + F.linear(F.gelu(F.linear(y, W[q + "mlp.fc1.weight"], W[q + "mlp.fc1.bias"])), W[q + "mlp.fc2.weight"], W[q + "mlp.fc2.bias"])
        if s < len(depths) - 1:                                      # SwinPatchMerging
            d = f"{p}encoder.layers.{s}.downsample."
            g = x.view(B, Hh, Ww, C)
            g = torch.cat([g[:, r::2, c::2, :] for c in range(2) for r in range(2)], -1).view(B, -1, 4 * C)
            g = F.layer_norm(g, (4 * C,), W[d + "norm.weight"], W[d + "norm.bias"], eps)
            x = F.linear(g, W[d + "reduction.weight"])
            Hh, Ww, C = Hh // 2, Ww // 2, 2 * C
    k = len(depths)
    x = F.layer_norm(x, (C,), W[prefix + f"hidden_states_norms.stage{k}.weight"], W[prefix + f"hidden_states_norms.stage{k}.bias"], eps)
    fm = x.view(B, Hh, Ww, C).permute(0, 3, 1, 2)
    return F.interpolate(fm, size=(out_hw, out_hw), mode="bilinear", align_corners=False)


# ----------------------------------------------------------------------------------------------
# embedding losses  (base_ola_vlm.py:289-320 ; ola_utils.py:96-125)
# ----------------------------------------------------------------------------------------------
def contrastive_loss(preds, targets, logit_scale, rank=0, gathered_targets=None):
    """calculate_contrastive_loss (ola_utils.py:108-125). `gathered_targets` = rank-ordered
    concatenation of every rank's L2-normalised targets (dist_collect :96-106)."""
    Bn = preds.shape[0]
    labels = torch.arange(Bn) + Bn * rank
    p = F.normalize(preds.flatten(1), dim=-1)
    t = F.normalize(targets.flatten(1), dim=-1)
    allt = t if gathered_targets is None else gathered_targets
    logits = p @ allt.t()
    sc = torch.clamp(logit_scale.exp(), max=100)
    return F.cross_entropy(logits * sc, labels, reduction="none")


def emb_loss(preds, mask, targets, logit_scale, w_contrastive, rank=0, gathered_targets=None):
    """_emb_loss (base_ola_vlm.py:289-320) incl. the outer-product mask broadcast."""
    targets = targets.to(preds.dtype)
    if targets.shape[0] != preds.shape[0]:                            # :292-299 (3-argument repeat: rank-3 targets only, like the reference)
        r = preds.shape[0] // targets.shape[0]
        targets = targets.repeat(r, 1, 1)
        mask = mask.repeat(r, 1, 1)
        if targets.shape[0] != preds.shape[0]:
            targets = targets[:preds.shape[0]]
            mask = mask[:preds.shape[0]]
    m = mask.view(preds.shape[0], *([1] * (preds.ndim - 1))).float()
    sl1 = F.smooth_l1_loss(preds.float(), targets.float(), reduction="none")
    con = contrastive_loss(preds, targets, logit_scale, rank, gathered_targets) if logit_scale is not None else 0
    sl1 = (sl1 * m).mean()
    con = (w_contrastive * con * m).mean()
    return sl1 + con, sl1, con


# ----------------------------------------------------------------------------------------------
# whole forward (ola_llama.py:79-188)
# ----------------------------------------------------------------------------------------------
def forward(W, batch, cfg, rank=0, gathered=None, need_logits=True):
    """batch: input_ids, attention_mask, labels, images, {gen,depth,seg}_target, {gen,depth,seg}_mask.
    Returns dict(loss, text_loss, logits, per-task loss triples, embeddings, layer_states...)."""
    images = batch["images"]
    if isinstance(images, (list, tuple)) or images.dim() == 5:        # ola_arch.py:262-275, mm_patch_merge_type "flat"
        ims = [x.unsqueeze(0) if x.dim() == 3 else x for x in images] if isinstance(images, (list, tuple)) else list(images)
        enc = encode_images(torch.cat(ims, 0), W, cfg)
        feats = [f.flatten(0, 1) for f in torch.split(enc, [x.shape[0] for x in ims], dim=0)]
    else:
        feats = encode_images(images, W, cfg)
    pid, am, emb, labels = prepare_inputs_labels_for_multimodal(
        batch["input_ids"], batch.get("attention_mask"), batch.get("labels"), feats, W, cfg)
    # the training caller passes no position_ids, so the reference drops the ones it built (`if _position_ids is None: position_ids = None`,
    # ola_arch.py:439-440) and HF numbers EVERY row of the padded tensor 0..S-1 — padded rows included (they are real query rows whose states the
    # shorter samples' heads read: pinned by tests/golden/tiny_llama_ragged.npz)
    hidden, states = decoder_forward(emb, pid if batch.get("position_ids") is not None else None, am, W, cfg)
    logits, text_loss = ntp_loss(hidden, labels, W, cfg, keep_logits=need_logits)
    out = dict(text_loss=text_loss, logits=logits, labels=labels,
               inputs_embeds=emb, image_features=feats, hidden=hidden, layer_states=states,
               layer_losses={})
    total = text_loss
    ns = num_sys_tokens(cfg)
    modes = cfg.aux_mode.split("-")
    spec = {"depth": ("image_depth", "depth_layer_indices", "depth_loss_weight", "depth_logit_scale"),
            "seg": ("image_seg", "seg_layer_indices", "seg_loss_weight", "seg_logit_scale"),
            "gen": ("image_gen", "img_layer_indices", "img_loss_weight", "gen_logit_scale")}
    for task in ("depth", "seg", "gen"):                      # call order: ola_llama.py:139-141
        if task not in modes or states[0].shape[1] <= ns or not getattr(cfg, "aux_heads", True):
            continue
        cname, ikey, wkey, sname = spec[task]
        hcfg = getattr(cfg, cname)
        tloss = 0
        embs = []
        for i, idx in enumerate(layer_indices(hcfg[ikey])):
            pred, extra = head_forward(states[idx], task, i, W, cfg)
            embs.append(extra if extra is not None else pred)
            if task == "depth" and "da_v2_head.depth_head.projects.0.weight" in W:
                with torch.no_grad():                          # base_ola_vlm.py:462-470
                    fe = [f.detach() for f in extra]
                    if len(fe) == 1:                           # use_intermediate_depth False: da_v2_head([depth_feats[0]] * 4), base_ola_vlm.py:465
                        fe = fe * 4
                    out.setdefault("depth_preds", []).append(dpt_depth_pred(fe, W))
            tgt = batch.get(f"{task}_target")
            if tgt is None:
                continue
            mask = batch[f"{task}_mask"].float()
            if cfg.zero_masks:
                mask = torch.zeros_like(mask)                  # base_ola_vlm.py:472-473,498-499,525-526
            scale = W.get(sname) if cfg.use_contrastive else None
            g = None if gathered is None else gathered[task]
            l, s, c = emb_loss(pred, mask, tgt, scale, cfg.contrastive_loss_weight, rank, g)
            out["layer_losses"][(task, idx)] = (l, s, c)
            tloss = tloss + l * hcfg[wkey]
        out[f"{task}_loss"] = tloss
        out[f"{task}_embs"] = embs
    for task in ("seg", "depth", "gen"):                      # sum order: ola_llama.py:143-144
        total = total + out.get(f"{task}_loss", 0)
    out["loss"] = total
    return out
