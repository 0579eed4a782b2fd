"""State-dict manifest of the reference PT-stage model (names and shapes exactly as the reference modules
create them — SURVEY §5.4 — so reference checkpoints round-trip) and an on-device random initialiser.

Names follow OlaLlavaLlamaForCausalLM / OlaLlavaPhi3ForCausalLM (ola_llama.py / ola_phi3.py), OlaLlavaMetaModel
(ola_arch.py:45,67-94,127), init_heads (base_ola_vlm.py:104-168), TaskTokenResampler (resampler.py:167-200),
TaskToken{Gen,Depth}Head / OneFormerTaskTokenSegHead, HF CLIPVisionModel, and the frozen DPT decoder
(`da_v2_head.*`, base_ola_vlm.py:139-148; DAv2_Head('vitl'), da_v2_head.py:182-314) whenever "depth" is in aux_mode."""
from __future__ import annotations

from collections import OrderedDict

from .config import layer_indices


def _resampler(pfx, dim, emb, out_dim, hc, sh, own_latents=False):
    inner = hc["num_heads"] * hc["dim_head"]
    if own_latents:                                   # Resampler (resampler.py:120-165): learned queries instead of task tokens
        sh[pfx + "latents"] = (1, hc["num_tokens"], dim)
    sh[pfx + "proj_in.weight"] = (dim, emb); sh[pfx + "proj_in.bias"] = (dim,)
    sh[pfx + "proj_out.weight"] = (out_dim, dim); sh[pfx + "proj_out.bias"] = (out_dim,)
    sh[pfx + "norm_out.weight"] = (out_dim,); sh[pfx + "norm_out.bias"] = (out_dim,)
    for d in range(hc["depth"]):
        a, f = f"{pfx}layers.{d}.0.", f"{pfx}layers.{d}.1."
        for n in ("norm1", "norm2"):
            sh[a + n + ".weight"] = (dim,); sh[a + n + ".bias"] = (dim,)
        sh[a + "to_q.weight"] = (inner, dim); sh[a + "to_kv.weight"] = (2 * inner, dim); sh[a + "to_out.weight"] = (dim, inner)
        ff = int(dim * hc["ff_mult"])
        sh[f + "0.weight"] = (dim,); sh[f + "0.bias"] = (dim,)
        sh[f + "1.weight"] = (ff, dim); sh[f + "3.weight"] = (dim, ff)


def param_shapes(cfg, vit_nested=True, with_vit=True) -> "OrderedDict[str, tuple]":
    """{name: shape}.  vit_nested=True uses transformers==4.41.1 naming (`...vision_tower.vision_model.*`, the
    reference's pin); False uses the flattened 5.x naming the golden fixtures were generated under."""
    sh = OrderedDict()
    H, V, F = cfg.hidden_size, cfg.vocab_size, cfg.intermediate_size
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    order = cfg.token_order
    nt = cfg.num_task_tokens
    heads = order if getattr(cfg, "aux_heads", True) else []          # IFT-stage classes (llava_llama.py) carry task tokens but no heads
    if "gen" in heads and hasattr(cfg, "image_gen"):
        sh["gen_logit_scale"] = ()
    if "depth" in heads and hasattr(cfg, "image_depth"):
        sh["depth_logit_scale"] = ()
    if "seg" in heads and hasattr(cfg, "image_seg"):
        sh["seg_logit_scale"] = ()
    if "gen" in heads and hasattr(cfg, "image_gen"):
        hc = cfg.image_gen
        for i in range(len(layer_indices(hc["img_layer_indices"]))):
            _resampler(f"image_gen_heads.{i}.projector.", hc["output_dim"], H, hc["output_dim"], hc, sh, nt == 0)   # gen_head.py:20-29 / 48-57
    if "depth" in heads and hasattr(cfg, "image_depth"):
        hc = cfg.image_depth
        for i in range(len(layer_indices(hc["depth_layer_indices"]))):
            # TaskTokenDepthHead: dim = llm hidden (da_v2_head.py:427-436); the num_task_tokens == 0 DepthHead: dim = output_dim (:386-395)
            _resampler(f"image_depth_heads.{i}.projector.", H if nt > 0 else hc["output_dim"], H, hc["output_dim"], hc, sh, nt == 0)
            # linear_1..3 exist only with use_intermediate_depth (da_v2_head.py:437-442; the flag's default: base_ola_vlm.py:132)
            for j in ((1, 2, 3) if hc.get("use_intermediate_depth", True) else ()):
                p = f"image_depth_heads.{i}.linear_{j}."
                sh[p + "0.weight"] = (hc["output_dim"], hc["output_dim"]); sh[p + "0.bias"] = (hc["output_dim"],)
                sh[p + "2.weight"] = (hc["output_dim"], hc["output_dim"]); sh[p + "2.bias"] = (hc["output_dim"],)
    if "seg" in heads and hasattr(cfg, "image_seg"):
        hc = cfg.image_seg
        for i in range(len(layer_indices(hc["seg_layer_indices"]))):
            _resampler(f"image_seg_heads.{i}.projector.", hc["output_dim"], H, hc["output_dim"], hc, sh, nt == 0)    # oneformer_head.py:197-206 / 233-242
    if nt > 0:
        if "depth" in order:
            sh["model.special_depth_tokens"] = (cfg.image_depth["num_tokens"], H)
        if "seg" in order:
            sh["model.special_seg_tokens"] = (cfg.image_seg["num_tokens"], H)
        if "gen" in order:
            sh["model.special_gen_tokens"] = (nt, H)
    sh["model.embed_tokens.weight"] = (V, H)
    for l in range(cfg.num_hidden_layers):
        p = f"model.layers.{l}."
        if cfg.arch == "phi3":
            sh[p + "self_attn.o_proj.weight"] = (H, nh * hd)
            sh[p + "self_attn.qkv_proj.weight"] = ((nh + 2 * nkv) * hd, H)
            sh[p + "mlp.gate_up_proj.weight"] = (2 * F, H)
            sh[p + "mlp.down_proj.weight"] = (H, F)
        else:
            sh[p + "self_attn.q_proj.weight"] = (nh * hd, H)
            sh[p + "self_attn.k_proj.weight"] = (nkv * hd, H)
            sh[p + "self_attn.v_proj.weight"] = (nkv * hd, H)
            sh[p + "self_attn.o_proj.weight"] = (H, nh * hd)
            sh[p + "mlp.gate_proj.weight"] = (F, H)
            sh[p + "mlp.up_proj.weight"] = (F, H)
            sh[p + "mlp.down_proj.weight"] = (H, F)
        sh[p + "input_layernorm.weight"] = (H,)
        sh[p + "post_attention_layernorm.weight"] = (H,)
    sh["model.norm.weight"] = (H,)
    if with_vit and getattr(cfg, "is_convnext", False):
        vp = "model.vision_tower.vision_tower."                      # timm ConvNeXt trunk (clip_convnext_encoder.py:119)
        dims, depths = cfg.cnx_dims, cfg.cnx_depths
        sh[vp + "stem.0.weight"] = (dims[0], 3, 4, 4); sh[vp + "stem.0.bias"] = (dims[0],)
        sh[vp + "stem.1.weight"] = (dims[0],); sh[vp + "stem.1.bias"] = (dims[0],)
        for i, (C, dep) in enumerate(zip(dims, depths)):
            q = f"{vp}stages.{i}."
            if i > 0:
                sh[q + "downsample.0.weight"] = (dims[i - 1],); sh[q + "downsample.0.bias"] = (dims[i - 1],)
                sh[q + "downsample.1.weight"] = (C, dims[i - 1], 2, 2); sh[q + "downsample.1.bias"] = (C,)
            for j in range(dep):
                b = f"{q}blocks.{j}."
                sh[b + "gamma"] = (C,)
                sh[b + "conv_dw.weight"] = (C, 1, 7, 7); sh[b + "conv_dw.bias"] = (C,)
                sh[b + "norm.weight"] = (C,); sh[b + "norm.bias"] = (C,)
                sh[b + "mlp.fc1.weight"] = (4 * C, C); sh[b + "mlp.fc1.bias"] = (4 * C,)
                sh[b + "mlp.fc2.weight"] = (C, 4 * C); sh[b + "mlp.fc2.bias"] = (C,)
    elif with_vit:
        vp = "model.vision_tower.vision_tower." + ("vision_model." if vit_nested else "")
        C, I = cfg.vit_hidden, cfg.vit_inter
        g = cfg.vit_image // cfg.vit_patch
        sh[vp + "embeddings.class_embedding"] = (C,)
        sh[vp + "embeddings.patch_embedding.weight"] = (C, 3, cfg.vit_patch, cfg.vit_patch)
        sh[vp + "embeddings.position_embedding.weight"] = (g * g + 1, C)
        sh[vp + "pre_layrnorm.weight"] = (C,); sh[vp + "pre_layrnorm.bias"] = (C,)
        for l in range(cfg.vit_layers):
            q = vp + f"encoder.layers.{l}."
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                sh[q + f"self_attn.{n}.weight"] = (C, C); sh[q + f"self_attn.{n}.bias"] = (C,)
            sh[q + "layer_norm1.weight"] = (C,); sh[q + "layer_norm1.bias"] = (C,)
            sh[q + "mlp.fc1.weight"] = (I, C); sh[q + "mlp.fc1.bias"] = (I,)
            sh[q + "mlp.fc2.weight"] = (C, I); sh[q + "mlp.fc2.bias"] = (C,)
            sh[q + "layer_norm2.weight"] = (C,); sh[q + "layer_norm2.bias"] = (C,)
        sh[vp + "post_layernorm.weight"] = (C,); sh[vp + "post_layernorm.bias"] = (C,)
    sh["model.mm_projector.0.weight"] = (H, cfg.mm_hidden_size); sh["model.mm_projector.0.bias"] = (H,)
    sh["model.mm_projector.2.weight"] = (H, H); sh["model.mm_projector.2.bias"] = (H,)
    sh["lm_head.weight"] = (V, H)
    if "depth" in heads and hasattr(cfg, "image_depth"):
        sh.update(dpt_param_shapes())
    return sh


DPT_OUT_CHANNELS = (256, 512, 1024, 1024)             # da_v2_head.py:300 (vitl)
DPT_FEATURES = 256


def dpt_param_shapes(prefix="da_v2_head.depth_head.") -> "OrderedDict[str, tuple]":
    """DAv2_Head('vitl').state_dict() (da_v2_head.py:182-258 DPTHead.__init__, :8-30 _make_scratch, :33-88 ResidualConvUnit,
    :91-125 FeatureFusionBlock), in module-registration order."""
    oc, f = DPT_OUT_CHANNELS, DPT_FEATURES
    sh = OrderedDict()
    for i, c in enumerate(oc):
        sh[prefix + f"projects.{i}.weight"] = (c, 1024, 1, 1); sh[prefix + f"projects.{i}.bias"] = (c,)
    sh[prefix + "resize_layers.0.weight"] = (oc[0], oc[0], 4, 4); sh[prefix + "resize_layers.0.bias"] = (oc[0],)
    sh[prefix + "resize_layers.1.weight"] = (oc[1], oc[1], 2, 2); sh[prefix + "resize_layers.1.bias"] = (oc[1],)
    sh[prefix + "resize_layers.3.weight"] = (oc[3], oc[3], 3, 3); sh[prefix + "resize_layers.3.bias"] = (oc[3],)
    sc = prefix + "scratch."
    for i, c in enumerate(oc):
        sh[sc + f"layer{i + 1}_rn.weight"] = (f, c, 3, 3)
    for r in (1, 2, 3, 4):
        q = sc + f"refinenet{r}."
        sh[q + "out_conv.weight"] = (f, f, 1, 1); sh[q + "out_conv.bias"] = (f,)
        for u in (1, 2):
            for cv in (1, 2):
                sh[q + f"resConfUnit{u}.conv{cv}.weight"] = (f, f, 3, 3); sh[q + f"resConfUnit{u}.conv{cv}.bias"] = (f,)
    sh[sc + "output_conv1.weight"] = (f // 2, f, 3, 3); sh[sc + "output_conv1.bias"] = (f // 2,)
    sh[sc + "output_conv2.0.weight"] = (32, f // 2, 3, 3); sh[sc + "output_conv2.0.bias"] = (32,)
    sh[sc + "output_conv2.2.weight"] = (1, 32, 1, 1); sh[sc + "output_conv2.2.bias"] = (1,)
    return sh


def init_value(name, shape, gen, device, dtype):
    """Random init mirroring the reference's module defaults closely enough for a throughput run:
    logit scales 2.0 (base_ola_vlm.py:113), task tokens ~N(0,1) (ola_arch.py:80-94), norm weights 1 / biases 0,
    linear & embedding weights ~N(0, 0.02) (HF initializer_range)."""
    import torch
    if name.endswith("logit_scale"):
        return torch.full((), 2.0, device=device, dtype=dtype)
    if "special_" in name:
        return torch.randn(shape, device=device, dtype=dtype, generator=gen)
    if name.endswith("projector.latents"):            # resampler.py:135: randn(1, num_queries, dim) / dim ** 0.5
        return torch.randn(shape, device=device, dtype=dtype, generator=gen) / shape[-1] ** 0.5
    if name.startswith("da_v2_head.") and len(shape) == 4:
        fan = shape[0] if ("resize_layers.0." in name or "resize_layers.1." in name) else shape[1] * shape[2] * shape[3]
        return torch.randn(shape, device=device, dtype=dtype, generator=gen) * (1.4 / fan ** 0.5)
    if len(shape) == 1:
        if name.endswith("bias"):
            return torch.zeros(shape, device=device, dtype=dtype)
        if name.endswith("class_embedding") or name.endswith(".gamma"):
            return torch.randn(shape, device=device, dtype=dtype, generator=gen) * 0.02
        return torch.ones(shape, device=device, dtype=dtype)
    return torch.randn(shape, device=device, dtype=dtype, generator=gen) * 0.02
