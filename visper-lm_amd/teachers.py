"""Frozen target encoders on the GPU.  The reference recomputes the distillation targets inside every training step,
one PIL image at a time (`base_ola_vlm.py:323-397`); here the depth teacher — DepthAnythingV2's DINOv2 ViT-L/14 backbone, target =
mean of the final-normed patch tokens of blocks [4, 11, 17, 23] (`base_ola_vlm.py:347-365`, `depth_anything_v2/dpt.py:164-169`,
`depth_anything_v2/dinov2.py:177-330`) — runs batched on the same kernels as the CLIP tower: im2col + GEMM(+position residual),
LayerNorm (eps 1e-6), fused-QKV GEMM + bias, non-causal flash attention (D = 64), out-proj / fc2 GEMMs with the LayerScale gammas
folded into the frozen weights and the residual add in the epilogue, fc1 GEMM + erf-GELU.  No gradient path.  `ClipImageEmbedTeacher` is the
generation teacher (unCLIP's CLIP ViT-H image encoder).  The OneFormer Swin-L teacher is still an input (`seg_target`)."""
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
        for l in range(self.L):
            q, o = f"{p}encoder.layers.{l}.", f"{l}."
            fz[o + "wqkv"] = d(torch.cat([W[q + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0))
            fz[o + "bqkv"] = d(torch.cat([W[q + f"self_attn.{x}_proj.bias"] for x in "qkv"], 0))
            fz[o + "wo"], fz[o + "bo"] = d(W[q + "self_attn.out_proj.weight"]), d(W[q + "self_attn.out_proj.bias"])
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
        hd = C // nh
        for l in range(self.L):
            o = f"{l}."
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln1w"], fz[o + "ln1b"], self.eps, save_stats=False)
            qkv = ops.gemm(y, fz[o + "wqkv"], bias=fz[o + "bqkv"]).view(B, N, 3 * C)
            att, _ = ops.attn_fwd(qkv[..., :C].view(B, N, nh, hd), qkv[..., C:2 * C].view(B, N, nh, hd), qkv[..., 2 * C:].view(B, N, nh, hd),
                                  causal=False)
            x = ops.gemm(att.view(B * N, C), fz[o + "wo"], bias=fz[o + "bo"], residual=x)
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln2w"], fz[o + "ln2b"], self.eps, save_stats=False)
            y = ops.gemm(y, fz[o + "w1"], bias=fz[o + "b1"], epi=self.epi)
            x = ops.gemm(y, fz[o + "w2"], bias=fz[o + "b2"], residual=x)
        cls = x.view(B, N, C)[:, 0].contiguous()
        pooled, _, _ = ops.layernorm_fwd(cls, fz["post_layernorm.w"], fz["post_layernorm.b"], self.eps, save_stats=False)
        return ops.gemm(pooled, fz["proj"])[:, :self.proj_dim].contiguous().view(B, 1, self.proj_dim)
