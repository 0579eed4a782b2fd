"""Frozen target encoders on the GPU.  The reference recomputes the distillation targets inside every training step,
one PIL image at a time (`base_ola_vlm.py:323-397`); here the depth teacher — DepthAnythingV2's DINOv2 ViT-L/14 backbone, target =
mean of the final-normed patch tokens of blocks [4, 11, 17, 23] (`base_ola_vlm.py:347-365`, `depth_anything_v2/dpt.py:164-169`,
`depth_anything_v2/dinov2.py:177-330`) — runs batched on the same kernels as the CLIP tower: im2col + GEMM(+position residual),
LayerNorm (eps 1e-6), fused-QKV GEMM + bias, non-causal flash attention (D = 64), out-proj / fc2 GEMMs with the LayerScale gammas
folded into the frozen weights and the residual add in the epilogue, fc1 GEMM + erf-GELU.  No gradient path.  `ClipImageEmbedTeacher` is the
generation teacher (unCLIP's CLIP ViT-H image encoder), `SwinSegTeacher` the segmentation teacher (OneFormer's Swin-L backbone)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ops

BF16 = torch.bfloat16
VITL_TAPS = (4, 11, 17, 23)                                   # dpt.py:164-169


class DinoV2DepthTeacher:
    def __init__(self, embed_dim=1024, depth=24, num_heads=16, taps=VITL_TAPS, patch=14, image=336, device="cuda",
                 prefix="dav2_backbone.pretrained."):
        if not torch.cuda.is_available():
            raise RuntimeError("DinoV2DepthTeacher needs a HIP device: there is no CPU fallback path")
        self.C, self.L, self.nh, self.taps, self.P, self.S = embed_dim, depth, num_heads, tuple(taps), patch, image
        self.dev, self.prefix, self.fz = torch.device(device), prefix, None

    @staticmethod
    def shapes(embed_dim=1024, depth=24, pos_grid=37, patch=14, prefix="dav2_backbone.pretrained."):
        """State-dict names / shapes of `DINOv2('vitl')` (dinov2.py:100-168, dinov2_layers/{block,attention,mlp,layer_scale}.py)."""
        C = embed_dim
        sh = {prefix + "cls_token": (1, 1, C), prefix + "pos_embed": (1, 1 + pos_grid * pos_grid, C), prefix + "mask_token": (1, C),
              prefix + "patch_embed.proj.weight": (C, 3, patch, patch), prefix + "patch_embed.proj.bias": (C,),
              prefix + "norm.weight": (C,), prefix + "norm.bias": (C,)}
        for i in range(depth):
            b = f"{prefix}blocks.{i}."
            for n in ("norm1", "norm2"):
                sh[b + n + ".weight"] = (C,); sh[b + n + ".bias"] = (C,)
            sh[b + "attn.qkv.weight"] = (3 * C, C); sh[b + "attn.qkv.bias"] = (3 * C,)
            sh[b + "attn.proj.weight"] = (C, C); sh[b + "attn.proj.bias"] = (C,)
            sh[b + "ls1.gamma"] = (C,); sh[b + "ls2.gamma"] = (C,)
            sh[b + "mlp.fc1.weight"] = (4 * C, C); sh[b + "mlp.fc1.bias"] = (4 * C,)
            sh[b + "mlp.fc2.weight"] = (C, 4 * C); sh[b + "mlp.fc2.bias"] = (C,)
        return sh

    def load_weights(self, W):
        p, C, P, dev = self.prefix, self.C, self.P, self.dev
        d = lambda t: t.detach().to(device=dev, dtype=BF16).contiguous()
        fz = self.fz = {}
        pw = W[p + "patch_embed.proj.weight"].reshape(C, -1)
        kp = (pw.shape[1] + 63) // 64 * 64
        pwp = torch.zeros(C, kp, dtype=torch.float32)
        pwp[:, :pw.shape[1]] = pw.float().cpu()
        fz["patch_w"], fz["patch_b"] = d(pwp), d(W[p + "patch_embed.proj.bias"])
        # position table for this input size, interpolated once on the host exactly as interpolate_pos_encoding does per call
        g = self.S // P
        pe = W[p + "pos_embed"].detach().float().cpu()
        N = pe.shape[1] - 1
        if N != g * g:
            sq = int(math.sqrt(N))
            sc = float(g + 0.1) / math.sqrt(N)
            pp = F.interpolate(pe[:, 1:].reshape(1, sq, sq, C).permute(0, 3, 1, 2), scale_factor=(sc, sc), mode="bicubic", antialias=False)
            assert pp.shape[-1] == g and pp.shape[-2] == g
            pe = torch.cat([pe[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, C)], 1)
        fz["pos"] = d(pe[0, 1:])
        fz["cls_pos"] = d(W[p + "cls_token"].float().cpu().reshape(C) + pe[0, 0])
        fz["norm_w"], fz["norm_b"] = d(W[p + "norm.weight"]), d(W[p + "norm.bias"])
        for i in range(max(self.taps) + 1):
            b, o = f"{p}blocks.{i}.", f"{i}."
            g1, g2 = W[b + "ls1.gamma"].float(), W[b + "ls2.gamma"].float()
            fz[o + "ln1w"], fz[o + "ln1b"] = d(W[b + "norm1.weight"]), d(W[b + "norm1.bias"])
            fz[o + "ln2w"], fz[o + "ln2b"] = d(W[b + "norm2.weight"]), d(W[b + "norm2.bias"])
            fz[o + "wqkv"], fz[o + "bqkv"] = d(W[b + "attn.qkv.weight"]), d(W[b + "attn.qkv.bias"])
            fz[o + "wo"], fz[o + "bo"] = d(W[b + "attn.proj.weight"].float() * g1[:, None]), d(W[b + "attn.proj.bias"].float() * g1)
            fz[o + "w1"], fz[o + "b1"] = d(W[b + "mlp.fc1.weight"]), d(W[b + "mlp.fc1.bias"])
            fz[o + "w2"], fz[o + "b2"] = d(W[b + "mlp.fc2.weight"].float() * g2[:, None]), d(W[b + "mlp.fc2.bias"].float() * g2)

    @torch.no_grad()
    def forward(self, images):
        """images [B, 3, S, S] (ImageNet-normalised, dpt.py:189-201) -> depth target [B, (S/14)^2, C] bf16."""
        fz, C, P, nh = self.fz, self.C, self.P, self.nh
        if fz is None:
            raise RuntimeError("load_weights() first")
        B = images.shape[0]
        g = self.S // P
        N = g * g + 1
        cols = images.to(device=self.dev, dtype=BF16).view(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * P * P)
        a = torch.zeros(B * g * g, fz["patch_w"].shape[1], device=self.dev, dtype=BF16)
        a[:, :3 * P * P] = cols
        h = torch.empty(B, N, C, device=self.dev, dtype=BF16)
        for b in range(B):
            ops.gemm(a[b * g * g:(b + 1) * g * g], fz["patch_w"], bias=fz["patch_b"], residual=fz["pos"], out=h[b, 1:])
        h[:, 0] = fz["cls_pos"]
        x = h.view(B * N, C)
        hd = C // nh
        acc = torch.zeros(B, g * g, C, device=self.dev, dtype=torch.float32)
        for i in range(max(self.taps) + 1):
            o = f"{i}."
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln1w"], fz[o + "ln1b"], 1e-6, save_stats=False)
            qkv = ops.gemm(y, fz[o + "wqkv"], bias=fz[o + "bqkv"]).view(B, N, 3 * C)
            att, _ = ops.attn_fwd(qkv[..., :C].view(B, N, nh, hd), qkv[..., C:2 * C].view(B, N, nh, hd), qkv[..., 2 * C:].view(B, N, nh, hd),
                                  causal=False)
            x = ops.gemm(att.view(B * N, C), fz[o + "wo"], bias=fz[o + "bo"], residual=x)          # x + gamma1 * proj(attn)
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln2w"], fz[o + "ln2b"], 1e-6, save_stats=False)
            y = ops.gemm(y, fz[o + "w1"], bias=fz[o + "b1"], epi=ops.EPI_GELU)
            x = ops.gemm(y, fz[o + "w2"], bias=fz[o + "b2"], residual=x)                          # x + gamma2 * fc2(gelu(fc1))
            if i in self.taps:
                t, _, _ = ops.layernorm_fwd(x, fz["norm_w"], fz["norm_b"], 1e-6, save_stats=False)
                ops.cast_to_f32(t.view(B, N, C)[:, 1:].contiguous(), out=acc, accumulate=True)
        return ops.cast_to_bf16(acc * (1.0 / len(self.taps))).view(B, g * g, C)


class ClipImageEmbedTeacher:
    """The generation teacher: `pipe.image_encoder(x).image_embeds` of the unCLIP pipeline (`base_ola_vlm.py:323-332`), i.e. HF
    `CLIPVisionModelWithProjection` (ViT-H/14 at 224 px: 1280-d, 32 layers, 16 heads, MLP 5120, erf-GELU, projection 1024): full tower,
    `post_layernorm` on the CLS row, bias-free `visual_projection` -> [B, 1, proj].  Batched on the CLIP-tower kernels; no gradient path."""

    def __init__(self, hidden=1280, layers=32, heads=16, image=224, patch=14, act="gelu", eps=1e-5, device="cuda", prefix="pipe.image_encoder."):
        if not torch.cuda.is_available():
            raise RuntimeError("ClipImageEmbedTeacher needs a HIP device: there is no CPU fallback path")
        self.C, self.L, self.nh, self.S, self.P, self.eps = hidden, layers, heads, image, patch, eps
        self.epi = ops.EPI_QUICK_GELU if act == "quick_gelu" else ops.EPI_GELU
        self.dev, self.prefix, self.fz = torch.device(device), prefix, None

    def load_weights(self, W):
        p, C, dev = self.prefix + "vision_model.", self.C, self.dev
        d = lambda t: t.detach().to(device=dev, dtype=BF16).contiguous()
        fz = self.fz = {}
        pw = W[p + "embeddings.patch_embedding.weight"].reshape(C, -1)
        kp = (pw.shape[1] + 63) // 64 * 64
        pwp = torch.zeros(C, kp, dtype=torch.float32)
        pwp[:, :pw.shape[1]] = pw.float().cpu()
        pos = W[p + "embeddings.position_embedding.weight"].float().cpu()
        fz["patch_w"], fz["pos"] = d(pwp), d(pos[1:])
        fz["cls_pos"] = d(W[p + "embeddings.class_embedding"].float().cpu() + pos[0])
        for n in ("pre_layrnorm", "post_layernorm"):
            fz[n + ".w"], fz[n + ".b"] = d(W[p + n + ".weight"]), d(W[p + n + ".bias"])
        # head_dim 80 (ViT-H) is not a kernel head size: every head is zero-padded to the next supported width INSIDE the frozen weights
        # (zero q/k/v rows, zero out-proj columns), so the padded features contribute exactly 0 and cost no extra pass
        nh, hd = self.nh, C // self.nh
        self.hp = hp = next(w for w in (32, 64, 96, 128) if w >= hd)

        def pad_rows(w):                                           # [nh*hd, ...] -> [nh*hp, ...]
            w = w.float().cpu()
            out = torch.zeros(nh, hp, *w.shape[1:])
            out[:, :hd] = w.view(nh, hd, *w.shape[1:])
            return out.view(nh * hp, *w.shape[1:])
        for l in range(self.L):
            q, o = f"{p}encoder.layers.{l}.", f"{l}."
            fz[o + "wqkv"] = d(torch.cat([pad_rows(W[q + f"self_attn.{x}_proj.weight"]) for x in "qkv"], 0))
            fz[o + "bqkv"] = d(torch.cat([pad_rows(W[q + f"self_attn.{x}_proj.bias"]) for x in "qkv"], 0))
            fz[o + "wo"] = d(pad_rows(W[q + "self_attn.out_proj.weight"].t().contiguous()).t().contiguous())      # zero columns
            fz[o + "bo"] = d(W[q + "self_attn.out_proj.bias"])
            for a, b in (("ln1", "layer_norm1"), ("ln2", "layer_norm2")):
                fz[o + a + "w"], fz[o + a + "b"] = d(W[q + b + ".weight"]), d(W[q + b + ".bias"])
            fz[o + "w1"], fz[o + "b1"] = d(W[q + "mlp.fc1.weight"]), d(W[q + "mlp.fc1.bias"])
            fz[o + "w2"], fz[o + "b2"] = d(W[q + "mlp.fc2.weight"]), d(W[q + "mlp.fc2.bias"])
        vp = W[self.prefix + "visual_projection.weight"]
        rows = (vp.shape[0] + 7) // 8 * 8                          # keep the GEMM's output rows 16-byte aligned
        vpp = torch.zeros(rows, vp.shape[1], dtype=torch.float32)
        vpp[:vp.shape[0]] = vp.float().cpu()
        fz["proj"], self.proj_dim = d(vpp), vp.shape[0]

    @torch.no_grad()
    def forward(self, images):
        """images [B, 3, S, S] (CLIP-normalised) -> image_embeds [B, 1, proj_dim] bf16."""
        fz, C, P, nh = self.fz, self.C, self.P, self.nh
        if fz is None:
            raise RuntimeError("load_weights() first")
        B = images.shape[0]
        g = self.S // P
        N = g * g + 1
        cols = images.to(device=self.dev, dtype=BF16).view(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * P * P)
        a = torch.zeros(B * g * g, fz["patch_w"].shape[1], device=self.dev, dtype=BF16)
        a[:, :3 * P * P] = cols
        h = torch.empty(B, N, C, device=self.dev, dtype=BF16)
        for b in range(B):
            ops.gemm(a[b * g * g:(b + 1) * g * g], fz["patch_w"], residual=fz["pos"], out=h[b, 1:])
        h[:, 0] = fz["cls_pos"]
        x, _, _ = ops.layernorm_fwd(h.view(B * N, C), fz["pre_layrnorm.w"], fz["pre_layrnorm.b"], self.eps, save_stats=False)
        hp, Cp = self.hp, nh * self.hp
        scale = (C // nh) ** -0.5
        for l in range(self.L):
            o = f"{l}."
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln1w"], fz[o + "ln1b"], self.eps, save_stats=False)
            qkv = ops.gemm(y, fz[o + "wqkv"], bias=fz[o + "bqkv"]).view(B, N, 3 * Cp)
            att, _ = ops.attn_fwd(qkv[..., :Cp].view(B, N, nh, hp), qkv[..., Cp:2 * Cp].view(B, N, nh, hp), qkv[..., 2 * Cp:].view(B, N, nh, hp),
                                  causal=False, scale=scale)
            x = ops.gemm(att.view(B * N, Cp), fz[o + "wo"], bias=fz[o + "bo"], residual=x)
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln2w"], fz[o + "ln2b"], self.eps, save_stats=False)
            y = ops.gemm(y, fz[o + "w1"], bias=fz[o + "b1"], epi=self.epi)
            x = ops.gemm(y, fz[o + "w2"], bias=fz[o + "b2"], residual=x)
        cls = x.view(B, N, C)[:, 0].contiguous()
        pooled, _, _ = ops.layernorm_fwd(cls, fz["post_layernorm.w"], fz["post_layernorm.b"], self.eps, save_stats=False)
        return ops.gemm(pooled, fz["proj"])[:, :self.proj_dim].contiguous().view(B, 1, self.proj_dim)


class SwinSegTeacher:
    """The segmentation teacher: OneFormer's Swin backbone, last feature map -> 24 x 24 (`base_ola_vlm.py:382-397` ->
    `oneformer_head.py:11-69`; HF modeling_swin.py).  Swin-L: embed 192, depths (2,2,18,2), heads (6,12,24,48) (head_dim 32), window 12,
    768-px input -> token grids 192 / 96 / 48 / 24.  Per block: LayerNorm -> cyclic shift + window partition (pure data movement) -> fused
    QKV GEMM -> window attention with the relative-position bias per head and the shift mask per window (`vp_attn_fwd_bias`, D = 32) ->
    out-proj GEMM -> window reverse + un-shift -> residual; LayerNorm -> fc1 GEMM + erf-GELU -> fc2 GEMM + residual.  Patch merging =
    2x2 gather + LayerNorm + bias-free GEMM.  Batched, no gradient path."""

    def __init__(self, embed_dim=192, depths=(2, 2, 18, 2), heads=(6, 12, 24, 48), window=12, patch=4, image=768, out_hw=24, eps=1e-5,
                 device="cuda", prefix="oneformer.model.pixel_level_module.encoder."):
        if not torch.cuda.is_available():
            raise RuntimeError("SwinSegTeacher needs a HIP device: there is no CPU fallback path")
        self.C0, self.depths, self.heads, self.ws, self.P, self.S, self.out_hw, self.eps = embed_dim, tuple(depths), tuple(heads), window, patch, image, out_hw, eps
        self.dev, self.prefix, self.fz = torch.device(device), prefix, None

    def load_weights(self, W):
        p, dev, ws = self.prefix + "swin.", self.dev, self.ws
        d = lambda t: t.detach().to(device=dev, dtype=BF16).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        fz = self.fz = {}
        pw = W[p + "embeddings.patch_embeddings.projection.weight"]
        pw = pw.reshape(pw.shape[0], -1)
        kp = (pw.shape[1] + 63) // 64 * 64
        pwp = torch.zeros(pw.shape[0], kp, dtype=torch.float32)
        pwp[:, :pw.shape[1]] = pw.float().cpu()
        fz["patch_w"], fz["patch_b"] = d(pwp), d(W[p + "embeddings.patch_embeddings.projection.bias"])
        fz["emb_ln_w"], fz["emb_ln_b"] = d(W[p + "embeddings.norm.weight"]), d(W[p + "embeddings.norm.bias"])
        c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1; rel[:, :, 1] += ws - 1; rel[:, :, 0] *= 2 * ws - 1
        ridx = rel.sum(-1).view(-1)
        N = ws * ws
        g = self.S // self.P
        for s, (dep, nh) in enumerate(zip(self.depths, self.heads)):
            for bi in range(dep):
                q, o = f"{p}encoder.layers.{s}.blocks.{bi}.", f"{s}.{bi}."
                fz[o + "ln1w"], fz[o + "ln1b"] = d(W[q + "layernorm_before.weight"]), d(W[q + "layernorm_before.bias"])
                fz[o + "ln2w"], fz[o + "ln2b"] = d(W[q + "layernorm_after.weight"]), d(W[q + "layernorm_after.bias"])
                fz[o + "wqkv"] = d(torch.cat([W[q + f"attention.{x}_proj.weight"] for x in "qkv"], 0))
                fz[o + "bqkv"] = d(torch.cat([W[q + f"attention.{x}_proj.bias"] for x in "qkv"], 0))
                fz[o + "wo"], fz[o + "bo"] = d(W[q + "attention.o_proj.weight"]), d(W[q + "attention.o_proj.bias"])
                tab = W[q + "attention.relative_position_bias.relative_position_bias_table"].float().cpu()
                fz[o + "bias"] = f32(tab[ridx].view(N, N, nh).permute(2, 0, 1))                       # [heads, N, N]
                fz[o + "w1"], fz[o + "b1"] = d(W[q + "mlp.fc1.weight"]), d(W[q + "mlp.fc1.bias"])
                fz[o + "w2"], fz[o + "b2"] = d(W[q + "mlp.fc2.weight"]), d(W[q + "mlp.fc2.bias"])
            # cyclic-shift mask of this stage's grid (SwinLayer.get_attn_mask): -100 between different shift regions
            sh = ws // 2
            hr = (torch.arange(g) >= g - ws).long() + (torch.arange(g) >= g - sh).long()
            img = (hr[:, None] * 3 + hr[None, :]).float()
            mw = img.view(g // ws, ws, g // ws, ws).transpose(1, 2).reshape(-1, N)
            dm = mw[:, None, :] - mw[:, :, None]
            fz[f"{s}.mask"] = f32(torch.where(dm != 0, torch.full_like(dm, -100.0), torch.zeros_like(dm)))
            if s < len(self.depths) - 1:
                dn = f"{p}encoder.layers.{s}.downsample."
                fz[f"{s}.dn_ln_w"], fz[f"{s}.dn_ln_b"] = d(W[dn + "norm.weight"]), d(W[dn + "norm.bias"])
                fz[f"{s}.dn_w"] = d(W[dn + "reduction.weight"])
                g //= 2
        k = len(self.depths)
        fz["out_ln_w"], fz["out_ln_b"] = d(W[self.prefix + f"hidden_states_norms.stage{k}.weight"]), d(W[self.prefix + f"hidden_states_norms.stage{k}.bias"])

    @torch.no_grad()
    def forward(self, images):
        """images [B, 3, S, S] (processor-normalised) -> seg target [B, C_last, 24, 24] bf16 (the layout `_get_seg_targets` returns)."""
        fz, ws, P = self.fz, self.ws, self.P
        if fz is None:
            raise RuntimeError("load_weights() first")
        B = images.shape[0]
        g = self.S // P
        C = self.C0
        N = ws * ws
        cols = images.to(device=self.dev, dtype=BF16).view(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * P * P)
        a = torch.zeros(B * g * g, fz["patch_w"].shape[1], device=self.dev, dtype=BF16)
        a[:, :3 * P * P] = cols
        x = ops.gemm(a, fz["patch_w"], bias=fz["patch_b"])
        x, _, _ = ops.layernorm_fwd(x, fz["emb_ln_w"], fz["emb_ln_b"], self.eps, save_stats=False)
        for s, (dep, nh) in enumerate(zip(self.depths, self.heads)):
            hd = C // nh
            nw = g // ws
            for bi in range(dep):
                o = f"{s}.{bi}."
                shift = 0 if bi % 2 == 0 else ws // 2
                y, _, _ = ops.layernorm_fwd(x, fz[o + "ln1w"], fz[o + "ln1b"], self.eps, save_stats=False)
                y = y.view(B, g, g, C)
                if shift:
                    y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
                win = y.view(B, nw, ws, nw, ws, C).transpose(2, 3).reshape(B * nw * nw * N, C)            # window partition
                qkv = ops.gemm(win, fz[o + "wqkv"], bias=fz[o + "bqkv"]).view(B * nw * nw, N, 3 * C)
                att = ops.attn_fwd_bias(qkv[..., :C].view(-1, N, nh, hd), qkv[..., C:2 * C].view(-1, N, nh, hd), qkv[..., 2 * C:].view(-1, N, nh, hd),
                                        bias_h=fz[o + "bias"], bias_b=fz[f"{s}.mask"] if shift else None)
                po = ops.gemm(att.view(-1, C), fz[o + "wo"], bias=fz[o + "bo"])
                po = po.view(B, nw, nw, ws, ws, C).transpose(2, 3).reshape(B, g, g, C)                     # window reverse
                if shift:
                    po = torch.roll(po, shifts=(shift, shift), dims=(1, 2))
                x = ops.add(x, po.reshape(B * g * g, C).contiguous())
                y, _, _ = ops.layernorm_fwd(x, fz[o + "ln2w"], fz[o + "ln2b"], self.eps, save_stats=False)
                y = ops.gemm(y, fz[o + "w1"], bias=fz[o + "b1"], epi=ops.EPI_GELU)
                x = ops.gemm(y, fz[o + "w2"], bias=fz[o + "b2"], residual=x)
            if s < len(self.depths) - 1:                                                                    # patch merging
                t = x.view(B, g, g, C)
                t = torch.cat([t[:, r::2, c::2, :] for c in range(2) for r in range(2)], -1).reshape(B * (g // 2) * (g // 2), 4 * C)
                t, _, _ = ops.layernorm_fwd(t.contiguous(), fz[f"{s}.dn_ln_w"], fz[f"{s}.dn_ln_b"], self.eps, save_stats=False)
                x = ops.gemm(t, fz[f"{s}.dn_w"])
                g, C = g // 2, 2 * C
        x, _, _ = ops.layernorm_fwd(x, fz["out_ln_w"], fz["out_ln_b"], self.eps, save_stats=False)
        fm = x.view(B, g, g, C)
        if g != self.out_hw:
            fm = ops.bilinear_nhwc(fm, self.out_hw, self.out_hw, align_corners=False)
        return fm.permute(0, 3, 1, 2).contiguous()
