"""The MI355X-native VisPer-LM pre-training step (NTP + per-layer embedding distillation).

`Engine` owns device-resident weights laid out for the HIP kernels and runs one fused forward+backward
of the reference's hot path entirely through the C ABI (ops.py -> libvisper_hip.so):

  CLIP-ViT tower (frozen, fwd only)                                   clip_encoder.py:47-59
  -> mlp2x_gelu projector (trainable)                                 ola_arch.py:187-190, builder.py:53-60
  -> token/image/task-token splice                                    ola_arch.py:224-254, 256-444
  -> Llama-3 / Phi-3 decoder, every layer state tapped on demand      ola_llama.py:105-119
  -> lm_head + shifted cross-entropy (row-chunked, logits never kept) ola_llama.py:121-136
  -> TaskTokenResampler heads + smooth-L1/InfoNCE losses              base_ola_vlm.py:289-320, 413-534
  -> backward: dgrad-only through the frozen decoder (PT stage: ola_vlm_train.py:1127-1131,1147),
     wgrad for projector / heads / task tokens / logit scales.

Data layout in HBM (sized for 288 GB): frozen decoder weights are kept twice — [out,in] for forward and
the pre-transposed [in,out] copy for dgrad — so both directions run the same K-contiguous MFMA GEMM;
q/k/v and gate/up are fused into single weights; activations needed by backward are kept (no recompute:
~1.5 GB/layer at B=8,S=2048).  Trainable parameters live in one flat fp32 master buffer (+ bf16 shadow,
fp32 grads, Adam moments) so the DP all-reduce and the fused AdamW are one launch each.
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .config import IGNORE_INDEX, IMAGE_TOKEN_INDEX, layer_indices, task_token_rows

BF16 = torch.bfloat16
F32 = torch.float32
from .splice import N_IMG_TOK

TASK_SPEC = {   # task -> (config attr, layer key, weight key, logit-scale name, heads module name)
    "depth": ("image_depth", "depth_layer_indices", "depth_loss_weight", "depth_logit_scale", "image_depth_heads"),
    "seg": ("image_seg", "seg_layer_indices", "seg_loss_weight", "seg_logit_scale", "image_seg_heads"),
    "gen": ("image_gen", "img_layer_indices", "img_loss_weight", "gen_logit_scale", "image_gen_heads"),
}


def is_trainable(name: str, train_llm: bool = False) -> bool:
    """PT-stage trainable set: projector, heads, task tokens, logit scales (ola_vlm_train.py:1127-1131, 1239-1242).
    IFT stage (`train_llm`, scripts/train/finetune.sh: full fine-tune, tower frozen): additionally every LLM parameter."""
    if "mm_projector" in name or "_heads." in name or "special_" in name or name.endswith("logit_scale"):
        return True
    return train_llm and not ("vision_tower" in name or name.startswith("da_v2_head."))


def _is_llm(name: str) -> bool:
    return name.startswith("model.layers.") or name in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")


class ParamStore:
    """Flat fp32 master / bf16 shadow / fp32 grad buffers for the trainable set (one all-reduce, one AdamW)."""

    def __init__(self, shapes: "OrderedDict[str, tuple]", device):
        self.index = OrderedDict()
        off = 0
        for name, shp in shapes.items():
            n = int(np.prod(shp)) if len(shp) else 1
            self.index[name] = (off, n, tuple(shp))
            off += (n + 63) // 64 * 64
        self.total = max(off, 64)
        self.master = torch.zeros(self.total, device=device, dtype=F32)
        self.shadow = torch.zeros(self.total, device=device, dtype=BF16)
        self.grad = torch.zeros(self.total, device=device, dtype=F32)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.step = 0
        self.version = 0          # bumped whenever the bf16 shadow changes (optimizer step, loads, external edits): keys derived caches

    def _v(self, buf, name):
        off, n, shp = self.index[name]
        return buf[off:off + n].view(shp if len(shp) else (1,))

    def w(self, name):        # bf16 kernel-side view
        return self._v(self.shadow, name)

    def p(self, name):        # fp32 master view
        return self._v(self.master, name)

    def g(self, name):        # fp32 grad view
        return self._v(self.grad, name)

    def __contains__(self, name):
        return name in self.index

    def refresh_shadow(self):
        ops.cast_to_bf16(self.master, out=self.shadow)
        self.version += 1

    def fused(self, names, buf=None):
        """View over parameters that sit back to back in the flat buffer (q|k|v, gate|up, to_q|to_kv) as ONE [sum(rows), cols] matrix."""
        off0, off = self.index[names[0]][0], self.index[names[0]][0]
        cols = self.index[names[0]][2][1]
        for n in names:
            o, cnt, shp = self.index[n]
            assert o == off and cnt % 64 == 0 and shp[1] == cols, f"{n} is not contiguous with its fusion partners"
            off += cnt
        return (self.shadow if buf is None else buf)[off0:off].view(-1, cols)

    def zero_grad(self):
        ops.zero_(self.grad)

    def grad_norm(self, grad_scale=1.0):
        """Global L2 norm of grad_scale * grad (host float; one small reduction kernel + a sync)."""
        return abs(grad_scale) * float(ops.sumsq(self.grad)) ** 0.5

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, mm_projector_lr=None,
                   lr_mult=1.0, max_grad_norm=None):
        """HF `adamw_torch` (ola_vlm_train.py:124) with the reference trainer's parameter groups (llava_trainer.py:890-995: no
        weight decay on biases / LayerNorm parameters, optional projector learning rate), `lr_mult` = the scheduler's multiplier
        for this step (optim.cosine_with_warmup) and optional global-norm clipping.  One fused launch per contiguous run of the
        flat buffer with equal (lr, weight decay) — a single launch in the reference configuration (weight_decay 0, one lr);
        refreshes the bf16 shadow in-kernel."""
        from . import optim
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.master)
            self.exp_avg_sq = torch.zeros_like(self.master)
        self.step += 1
        self.version += 1
        if max_grad_norm is not None:
            grad_scale = grad_scale * optim.clip_coefficient(self.grad_norm(grad_scale), max_grad_norm)
        key = (float(weight_decay), mm_projector_lr)
        if getattr(self, "_runs_key", None) != key:
            self._runs = optim.runs(self.index, optim.param_groups(self.index.keys(), weight_decay, mm_projector_lr))
            self._runs_key = key
        for a, b, glr, wd in self._runs:
            ops.adamw_(self.master[a:b], self.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.shadow[a:b],
                       (lr if glr is None else glr) * lr_mult, betas[0], betas[1], eps, wd, self.step, grad_scale)


def _tp(w):
    """Device transpose of a 2-D bf16 weight (pre-transposed dgrad copy)."""
    return ops.transpose(w.contiguous())


class LazyRandomWeights:
    """Mapping name -> freshly generated bf16 device tensor (random init, see params.init_value); lets an 8 B-parameter
    model be initialised layer by layer on the GPU without a host copy.  Deterministic for a given seed and access order."""

    def __init__(self, cfg, device, seed=0, vit_nested=True):
        from .params import param_shapes
        self.shapes = param_shapes(cfg, vit_nested=vit_nested)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def __contains__(self, k):
        return k in self.shapes

    def __getitem__(self, k):
        from .params import init_value
        return init_value(k, self.shapes[k], self.gen, self.device, BF16)


class Engine:
    def __init__(self, cfg, device="cuda", lm_chunk_rows=16384):
        if not torch.cuda.is_available():
            raise RuntimeError("visper_lm_amd.Engine needs a HIP device: there is no CPU fallback path")
        self.cfg = cfg
        self.dev = torch.device(device)
        self.lm_chunk_rows = lm_chunk_rows
        self.rank, self.world = 0, 1
        self.fz = {}           # frozen, kernel-ready weights
        self.ps = None         # ParamStore
        self._rope = {}
        self._plan_cache = {}          # per-batch splice plans (small LRU-ish cache: identical batches reuse their tables)
        self._static = {}              # shape-keyed device tables that never depend on the batch contents
        self._red = None
        self.comm = None       # parallel.NativeComm when the C ABI's own communicator carries the collectives (set_distributed)
        self.keep_logits = False
        self.lm_head_all_rows = False # True: no labelled-row compaction — every row through lm_head + CE + the d_hidden GEMM, literally as ola_llama.py:121-136 (A/B + test aid)
        self.keep_states = False      # True: train_step also returns every decoder-layer state (HF `hidden_states`: embeddings, layer outputs, final norm)

    def _load_frozen_decoder(self, W, d):
        """PT stage: the LLM is frozen -> bf16 kernel-ready copies, q/k/v and gate/up fused, plus pre-transposed dgrad copies."""
        cfg, fz = self.cfg, self.fz
        # RMSNorm fold (frozen LLM only; VP_FOLD_NORM=0 turns it off): y = W (gamma (.) x / rms) = (W diag(gamma)) x / rms.  The kernel-side decoder
        # weights are derived copies already (fused qkv, interleaved gate|up, pre-transposed dgrad copies), so gamma goes into them for free, the
        # norms keep gamma = 1, and where every GEMM of a layer is a one-wave-per-SIMD launch (ops.fold_norm_ok) the normalisation itself
        # disappears: 1/rms is a row scale in the consumer's epilogue, the sums of squares come out of the producer's residual epilogue.
        # Reference: HF LlamaRMSNorm / LlamaDecoderLayer (modeling_llama.py), reached from ola_llama.py:105-115.
        # Only where it pays: head_dim 128 (RoPE in the QKV epilogue as well).  Phi-3 (D = 96) was measured with the fold in round 5 (1/rms as the QKV
        # GEMM's row scale with rope_cols = 0, the rotation still its own kernel: a 96-wide head straddles the 128-column wave sub-tiles): 247.74 vs
        # 247.84 ms/step = nothing, so it keeps gamma in its norms and the reference's rounding points (the code path stays: VP_FOLD_NORM_96=1).
        self.fold_norm = os.environ.get("VP_FOLD_NORM", "1") != "0" and cfg.hidden_size % 256 == 0 and \
            (cfg.head_dim == 128 or (cfg.head_dim == 96 and os.environ.get("VP_FOLD_NORM_96") == "1"))
        fz["ones_h"] = torch.ones(cfg.hidden_size, device=self.dev, dtype=BF16)
        fz["embed"] = d(W["model.embed_tokens.weight"])
        fz["norm"] = d(W["model.norm.weight"])
        fz["lm_head"] = d(W["lm_head.weight"])
        fz["lm_head_T"] = _tp(fz["lm_head"])
        for l in range(cfg.num_hidden_layers):
            p = f"model.layers.{l}."
            o = f"dec.{l}."
            if cfg.arch == "phi3":
                wqkv, wgu = W[p + "self_attn.qkv_proj.weight"], W[p + "mlp.gate_up_proj.weight"]
            else:
                wqkv = torch.cat([W[p + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0)
                wgu = torch.cat([W[p + "mlp.gate_proj.weight"], W[p + "mlp.up_proj.weight"]], 0)
            if self.fold_norm:                                   # gamma of the two RMSNorms into the (derived, frozen) consumer weights: see _decoder_fwd
                g1, g2 = W[p + "input_layernorm.weight"], W[p + "post_attention_layernorm.weight"]
                wqkv = wqkv.to(self.dev).float() * g1.to(self.dev).float()[None, :]
                wgu = wgu.to(self.dev).float() * g2.to(self.dev).float()[None, :]
            fz[o + "wqkv"] = d(wqkv)
            fz[o + "wgu"] = ops.interleave_gate_up(d(wgu))      # gate / up rows interleaved in 8-row chunks (fused SwiGLU epilogues)
            fz[o + "wo"] = d(W[p + "self_attn.o_proj.weight"])
            fz[o + "wd"] = d(W[p + "mlp.down_proj.weight"])
            for k in ("wqkv", "wgu", "wo", "wd"):
                fz[o + k + "_T"] = _tp(fz[o + k])
            if self.fold_norm:
                fz[o + "ln1"] = fz[o + "ln2"] = fz["ones_h"]
            else:
                fz[o + "ln1"] = d(W[p + "input_layernorm.weight"])
                fz[o + "ln2"] = d(W[p + "post_attention_layernorm.weight"])

    # ------------------------------------------------------------------------------------------ weights
    def load_weights(self, W):
        """W: {reference state-dict name: tensor}. Frozen weights -> bf16 device buffers (fused / pre-transposed);
        trainable ones -> ParamStore."""
        cfg, dev = self.cfg, self.dev
        d = lambda t: t.detach().to(device=dev, dtype=BF16).contiguous()
        fz = self.fz = {}
        # ---- CLIP tower (both transformers naming generations: with / without ".vision_model")
        vp = "model.vision_tower.vision_tower.vision_model."
        if vp + "embeddings.class_embedding" not in W:
            vp = "model.vision_tower.vision_tower."
        self.vit_prefix = vp
        if vp + "embeddings.class_embedding" in W:
            C = cfg.vit_hidden
            pw = W[vp + "embeddings.patch_embedding.weight"].reshape(C, -1)
            kp = (pw.shape[1] + 63) // 64 * 64
            pwp = torch.zeros(C, kp, dtype=pw.dtype, device=pw.device)
            pwp[:, :pw.shape[1]] = pw
            fz["vit.patch_w"] = d(pwp)
            pos = W[vp + "embeddings.position_embedding.weight"]
            fz["vit.pos"] = d(pos[1:])
            fz["vit.cls_pos"] = d((W[vp + "embeddings.class_embedding"].to(BF16) + pos[0].to(BF16)))
            for nm in ("pre_layrnorm.weight", "pre_layrnorm.bias"):
                fz["vit." + nm] = d(W[vp + nm])
            for l in range(self.vit_layers_run()):
                q = vp + f"encoder.layers.{l}."
                o = f"vit.{l}."
                fz[o + "wqkv"] = d(torch.cat([W[q + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0))
                fz[o + "bqkv"] = d(torch.cat([W[q + f"self_attn.{x}_proj.bias"] for x in "qkv"], 0))
                for a, b in (("wo", "self_attn.out_proj.weight"), ("bo", "self_attn.out_proj.bias"),
                             ("ln1w", "layer_norm1.weight"), ("ln1b", "layer_norm1.bias"), ("ln2w", "layer_norm2.weight"),
                             ("ln2b", "layer_norm2.bias"), ("w1", "mlp.fc1.weight"), ("b1", "mlp.fc1.bias"),
                             ("w2", "mlp.fc2.weight"), ("b2", "mlp.fc2.bias")):
                    fz[o + a] = d(W[q + b])
        # ---- CLIP-ConvNeXt trunk (config 4): channels-last weights, gamma folded into fc2, conv weights as GEMM operands
        cp = "model.vision_tower.vision_tower."
        self.has_convnext = (cp + "stem.0.weight") in W
        if self.has_convnext:
            sw = W[cp + "stem.0.weight"].reshape(cfg.cnx_dims[0], -1)
            swp = torch.zeros(sw.shape[0], 64, dtype=sw.dtype, device=sw.device)
            swp[:, :sw.shape[1]] = sw
            fz["cnx.stem_w"], fz["cnx.stem_b"] = d(swp), d(W[cp + "stem.0.bias"])
            fz["cnx.stem_ln_w"], fz["cnx.stem_ln_b"] = d(W[cp + "stem.1.weight"]), d(W[cp + "stem.1.bias"])
            for i, dep in enumerate(cfg.cnx_depths):
                q = f"{cp}stages.{i}."
                if i > 0:
                    fz[f"cnx.{i}.ds_ln_w"], fz[f"cnx.{i}.ds_ln_b"] = d(W[q + "downsample.0.weight"]), d(W[q + "downsample.0.bias"])
                    dw_ = W[q + "downsample.1.weight"]                       # [Cout, Cin, 2, 2] -> [Cout, (dy,dx,cin)]
                    fz[f"cnx.{i}.ds_w"] = d(dw_.permute(0, 2, 3, 1).reshape(dw_.shape[0], -1))
                    fz[f"cnx.{i}.ds_b"] = d(W[q + "downsample.1.bias"])
                for j in range(dep):
                    b_, o = f"{q}blocks.{j}.", f"cnx.{i}.{j}."
                    C = W[b_ + "gamma"].shape[0]
                    fz[o + "dw_w"] = d(W[b_ + "conv_dw.weight"].reshape(C, 49).t())     # tap-major [49, C]
                    fz[o + "dw_b"] = d(W[b_ + "conv_dw.bias"])
                    fz[o + "ln_w"], fz[o + "ln_b"] = d(W[b_ + "norm.weight"]), d(W[b_ + "norm.bias"])
                    fz[o + "w1"], fz[o + "b1"] = d(W[b_ + "mlp.fc1.weight"]), d(W[b_ + "mlp.fc1.bias"])
                    gam = W[b_ + "gamma"].float()
                    fz[o + "w2"] = d(W[b_ + "mlp.fc2.weight"].float() * gam[:, None])      # layer scale folded (frozen tower)
                    fz[o + "b2"] = d(W[b_ + "mlp.fc2.bias"].float() * gam)
        # ---- decoder (frozen copies; with a trainable LLM the kernel-side weights alias the parameter store instead, see below)
        train_llm = bool(getattr(cfg, "train_llm", False))
        self.gu_interleaved = not train_llm
        if not train_llm:
            self._load_frozen_decoder(W, d)
        # ---- frozen DPT depth decoder (a11; da_v2_head.py:182-314): conv weights as GEMM matrices over NHWC activations
        dp = "da_v2_head.depth_head."
        if dp + "projects.0.weight" in W:
            c3 = lambda k: d(W[k].permute(0, 2, 3, 1).reshape(W[k].shape[0], -1))            # [Co,Ci,3,3] -> [Co, (ky,kx,ci)]
            c1 = lambda k: d(W[k].reshape(W[k].shape[0], -1))                                # [Co,Ci,1,1] -> [Co, Ci]
            for i in range(4):
                fz[f"dpt.proj{i}.w"], fz[f"dpt.proj{i}.b"] = c1(dp + f"projects.{i}.weight"), d(W[dp + f"projects.{i}.bias"])
                fz[f"dpt.rn{i}.w"] = c3(dp + f"scratch.layer{i + 1}_rn.weight")
            for i, k in ((0, 4), (1, 2)):                                                    # ConvTranspose2d(k = stride): [Ci,Co,k,k]
                w = W[dp + f"resize_layers.{i}.weight"]
                fz[f"dpt.up{i}.w"] = d(w.permute(2, 3, 1, 0).reshape(k * k * w.shape[1], w.shape[0]))     # rows (ky, kx, co)
                fz[f"dpt.up{i}.b"] = d(W[dp + f"resize_layers.{i}.bias"].repeat(k * k))
            fz["dpt.down3.w"], fz["dpt.down3.b"] = c3(dp + "resize_layers.3.weight"), d(W[dp + "resize_layers.3.bias"])
            for r in (1, 2, 3, 4):
                q = dp + f"scratch.refinenet{r}."
                fz[f"dpt.ref{r}.out.w"], fz[f"dpt.ref{r}.out.b"] = c1(q + "out_conv.weight"), d(W[q + "out_conv.bias"])
                for u in (1, 2):
                    for cv in (1, 2):
                        fz[f"dpt.ref{r}.rcu{u}.c{cv}.w"] = c3(q + f"resConfUnit{u}.conv{cv}.weight")
                        fz[f"dpt.ref{r}.rcu{u}.c{cv}.b"] = d(W[q + f"resConfUnit{u}.conv{cv}.bias"])
            sc = dp + "scratch."
            fz["dpt.oc1.w"], fz["dpt.oc1.b"] = c3(sc + "output_conv1.weight"), d(W[sc + "output_conv1.bias"])
            fz["dpt.oc2a.w"], fz["dpt.oc2a.b"] = c3(sc + "output_conv2.0.weight"), d(W[sc + "output_conv2.0.bias"])
            # the last 1x1 conv has one output channel: pad to 8 rows so the GEMM output rows stay 16-byte aligned
            w1 = torch.zeros(8, 32, dtype=torch.float32, device=W[sc + "output_conv2.2.weight"].device)
            w1[0] = W[sc + "output_conv2.2.weight"].reshape(-1).float()
            b1 = torch.zeros(8, dtype=torch.float32, device=w1.device)
            b1[0] = W[sc + "output_conv2.2.bias"].reshape(-1).float()[0]
            fz["dpt.oc2b.w"], fz["dpt.oc2b.b"] = d(w1), d(b1)
        # ---- trainable.  Flat-buffer order: [heads + logit scales | projector + task tokens] so the first block's
        # gradients (final as soon as the heads' backward is done) can be all-reduced under the decoder backward.
        allshapes = getattr(W, "shapes", None) or OrderedDict((k, tuple(v.shape)) for k, v in W.items())
        tr = [k for k in allshapes if is_trainable(k, train_llm)]
        early = [k for k in tr if ("_heads." in k or k.endswith("logit_scale"))]
        late = [k for k in tr if k not in set(early) and not _is_llm(k)]
        # IFT: LLM parameters follow in BACKWARD order (lm_head, final norm, layers L-1..0, embeddings) so that each gradient
        # bucket that becomes final during the backward pass is one contiguous range of the flat buffer
        llm = [k for k in tr if _is_llm(k)]
        lay = lambda k: int(k.split(".")[2]) if k.startswith("model.layers.") else -1
        llm_order = ([k for k in llm if k == "lm_head.weight"] + [k for k in llm if k == "model.norm.weight"] +
                     [k for l in range(cfg.num_hidden_layers - 1, -1, -1) for k in llm if lay(k) == l] +
                     [k for k in llm if k == "model.embed_tokens.weight"])
        shapes = OrderedDict((k, tuple(allshapes[k])) for k in early + late + llm_order)
        self.ps = ParamStore(shapes, dev)
        self.ps.split = self.ps.index[late[0]][0] if late else self.ps.total
        for k in shapes:
            self.ps.p(k).copy_(W[k].detach().to(device=dev, dtype=F32).reshape(self.ps.p(k).shape))
        self.ps.refresh_shadow()
        self.train_llm = train_llm
        if train_llm:
            self._alias_llm_weights()
        self._build_static()

    # ------------------------------------------------------------------------------------------ IFT: trainable LLM
    def _fused_range(self, names):
        """(offset, numel) of parameters that sit back to back in the flat buffer (q,k,v / gate,up)."""
        ps = self.ps
        off0 = ps.index[names[0]][0]
        off = off0
        for n in names:
            o, cnt, _ = ps.index[n]
            assert o == off and cnt % 64 == 0, f"{n} is not contiguous with its fusion partners"
            off += cnt
        return off0, off - off0

    def _alias_llm_weights(self):
        """Kernel-side decoder weights = views of the parameter store's bf16 shadow (updated in place by the fused AdamW); q/k/v and
        gate/up are adjacent in the flat buffer, so their fused matrices are plain views ([gate | up] halves, not interleaved: the
        names must stay contiguous views for state-dict I/O).  The [in,out] dgrad copies are re-transposed after every step."""
        cfg, fz, ps = self.cfg, self.fz, self.ps
        H = cfg.hidden_size
        self._llm_ranges = {}
        fz["embed"], fz["norm"], fz["lm_head"] = ps.w("model.embed_tokens.weight"), ps.w("model.norm.weight"), ps.w("lm_head.weight")
        fz["lm_head_T"] = torch.empty(H, cfg.vocab_size, device=self.dev, dtype=BF16)
        for l in range(cfg.num_hidden_layers):
            p, o = f"model.layers.{l}.", f"dec.{l}."
            if cfg.arch == "phi3":
                qn, gn = [p + "self_attn.qkv_proj.weight"], [p + "mlp.gate_up_proj.weight"]
            else:
                qn = [p + f"self_attn.{x}_proj.weight" for x in "qkv"]
                gn = [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"]
            (oq, nq), (og, ng) = self._fused_range(qn), self._fused_range(gn)
            fz[o + "wqkv"] = ps.shadow[oq:oq + nq].view(-1, H)
            fz[o + "wgu"] = ps.shadow[og:og + ng].view(-1, H)
            self._llm_ranges[o + "wqkv"], self._llm_ranges[o + "wgu"] = (oq, nq), (og, ng)
            fz[o + "wo"], fz[o + "wd"] = ps.w(p + "self_attn.o_proj.weight"), ps.w(p + "mlp.down_proj.weight")
            fz[o + "ln1"], fz[o + "ln2"] = ps.w(p + "input_layernorm.weight"), ps.w(p + "post_attention_layernorm.weight")
            for k in ("wqkv", "wgu", "wo", "wd"):
                w = fz[o + k]
                fz[o + k + "_T"] = torch.empty(w.shape[1], w.shape[0], device=self.dev, dtype=BF16)
            names = [n for n in ps.index if n.startswith(p)]
            lo = min(ps.index[n][0] for n in names)
            hi = max(ps.index[n][0] + (ps.index[n][1] + 63) // 64 * 64 for n in names)
            self._llm_ranges[o] = (lo, hi - lo)
        self.refresh_transposes()

    def refresh_transposes(self):
        """[in,out] copies of the (just updated) trainable decoder weights for the dgrad GEMMs."""
        fz = self.fz
        ops.transpose(fz["lm_head"], out=fz["lm_head_T"])
        for l in range(self.cfg.num_hidden_layers):
            o = f"dec.{l}."
            for k in ("wqkv", "wgu", "wo", "wd"):
                ops.transpose(fz[o + k], out=fz[o + k + "_T"])

    def _gfused(self, key):
        """fp32 gradient view of a fused weight (same layout as the fused bf16 view)."""
        off, n = self._llm_ranges[key]
        return self.ps.grad[off:off + n].view(-1, self.cfg.hidden_size)

    def _reduce_range(self, off, n):
        if self.world > 1:
            self._reducer().reduce_range(off, off + n)

    def init_random(self, seed=0):
        """Random-init weights of the configured architecture, generated on the device (bench / smoke runs)."""
        self.load_weights(LazyRandomWeights(self.cfg, self.dev, seed))

    def set_distributed(self, rank, world, transport=None):
        """transport "torch" (default): torch.distributed collectives (backend nccl = RCCL); "native": the C ABI's own communicator
        (vp_comm_*: RCCL + side stream + event fences inside libvisper_hip.so).  VP_COMM overrides the default."""
        self.rank, self.world = rank, world
        transport = transport or os.environ.get("VP_COMM", "torch")
        if transport not in ("torch", "native"):
            raise ValueError(f"transport={transport!r}")
        self.comm = None
        if transport == "native" and self.dev.type == "cuda":
            from .parallel import NativeComm
            self.comm = NativeComm(rank, world)
        self._red = None
        # with collectives running beside the GEMMs the persistent kernel claims its tiles dynamically (a CU held by an RCCL kernel then
        # costs its own share instead of stalling the whole static grid); single GPU keeps the static walk
        if self.dev.type == "cuda" and os.environ.get("VP_GEMM_DYN") is None:
            ops.set_dynamic(world > 1)

    def _reducer(self):
        from .parallel import GradReducer
        if self._red is None or self._red.g is not self.ps.grad:
            wire = torch.float32 if str(getattr(self.cfg, "grad_reduce_dtype", "bf16")) in ("fp32", "float32") else BF16
            self._red = GradReducer(self.ps.grad, self.ps.split, comm=getattr(self, "comm", None), reduce_dtype=wire)
        self._red.dry = bool(getattr(self, "comm_dry", False))
        return self._red

    def finish_grads(self):
        """Join outstanding gradient all-reduces (the compute stream waits; the host does not)."""
        if self.world > 1:
            self._reducer().finish()

    def optimizer_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, mm_projector_lr=None, lr_mult=1.0,
                       max_grad_norm=None):
        """Mean-reduce over DP ranks is folded into AdamW's grad_scale (ZeRO-2 / DDP averaging semantics); groups, schedule
        multiplier and clipping as in ParamStore.adamw_step."""
        self.finish_grads()
        self.ps.adamw_step(lr, betas, eps, weight_decay, grad_scale=1.0 / self.world, mm_projector_lr=mm_projector_lr,
                           lr_mult=lr_mult, max_grad_norm=max_grad_norm)
        if getattr(self, "train_llm", False):
            self.refresh_transposes()

    # ------------------------------------------------------------------------------------------ state I/O
    def state_dict(self, dtype=BF16):
        """{reference state-dict name: tensor} of every TRAINABLE parameter (fp32 master -> `dtype`, the reference's bf16 by default;
        logit scales stay fp32 like the reference's nn.Parameter(torch.tensor(2.0))): what HF `trainer.save_model` persists of the
        trained modules — projector, heads, special_*_tokens, *_logit_scale (PT), plus the LLM when it trains (IFT)."""
        out = OrderedDict()
        for n, (_, _, shp) in self.ps.index.items():
            t = self.ps.p(n).detach().reshape(shp)
            out[n] = t.clone() if len(shp) == 0 else t.to(dtype)
        return out

    def load_state_dict(self, sd, strict=True):
        """Load trainable parameters by reference name (a checkpoint of a previous stage / run); returns the names loaded."""
        ps = self.ps
        missing = [n for n in ps.index if n not in sd]
        if strict and missing:
            raise KeyError(f"missing trainable parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        loaded = []
        for n in ps.index:
            if n in sd:
                ps.p(n).copy_(sd[n].detach().to(device=self.dev, dtype=F32).reshape(ps.p(n).shape))
                loaded.append(n)
        ps.refresh_shadow()
        if getattr(self, "train_llm", False):
            self.refresh_transposes()
        return loaded

    def optimizer_state_dict(self):
        """AdamW state for exact resume: step counter + the flat fp32 master and moments (the fp32 master is part of the optimizer
        state exactly as in DeepSpeed's bf16 optimizer: the bf16 model weights alone do not reproduce the trajectory)."""
        ps = self.ps
        z = lambda t: None if t is None else t.detach().cpu().clone()
        return dict(step=ps.step, names=list(ps.index), offsets=[ps.index[n][0] for n in ps.index], total=ps.total,
                    master=z(ps.master), exp_avg=z(ps.exp_avg), exp_avg_sq=z(ps.exp_avg_sq))

    def load_optimizer_state_dict(self, st):
        ps = self.ps
        if list(st["names"]) != list(ps.index) or int(st["total"]) != ps.total:
            raise ValueError("optimizer state was saved for a different trainable set / layout")
        ps.step = int(st["step"])
        ps.master.copy_(st["master"].to(self.dev))
        for k in ("exp_avg", "exp_avg_sq"):
            setattr(ps, k, None if st[k] is None else st[k].to(device=self.dev, dtype=F32).clone())
        ps.refresh_shadow()
        if getattr(self, "train_llm", False):
            self.refresh_transposes()

    def vit_layers_run(self):
        sel = self.cfg.mm_vision_select_layer
        return self.cfg.vit_layers + 1 + sel if sel < 0 else sel

    def _build_static(self):
        cfg = self.cfg
        self.tasks = []          # [(task, head_i, layer_idx)]
        for task in ("depth", "seg", "gen"):                      # reference call order: ola_llama.py:139-141
            if task in cfg.token_order and hasattr(cfg, TASK_SPEC[task][0]) and getattr(cfg, "aux_heads", True):
                hc = getattr(cfg, TASK_SPEC[task][0])
                for i, idx in enumerate(layer_indices(hc[TASK_SPEC[task][1]])):
                    self.tasks.append((task, i, idx))
        self.tapped = sorted({idx for _, _, idx in self.tasks})

    def _zeros_i32(self, n):
        key = ("zeros_i32", n)
        if key not in self._static:
            self._static[key] = torch.zeros(n, device=self.dev, dtype=torch.int32)
        return self._static[key]

    def _arange(self, n):
        key = ("arange", n)
        if key not in self._static:
            self._static[key] = torch.arange(n, device=self.dev, dtype=torch.int32)
        return self._static[key]

    def rope(self, S):
        if S not in self._rope:
            self._rope[S] = ops.rope_tables(S, self.cfg.head_dim, self.cfg.rope_theta, self.dev)
        return self._rope[S]

    # ------------------------------------------------------------------------------------------ ViT
    def convnext_forward(self, images):
        """Frozen CLIP-ConvNeXt trunk (clip_convnext_encoder.py:150-174), channels-last end to end:
        stem 4x4/s4 as im2col GEMM + LN; per block: depthwise 7x7 kernel -> LN -> GEMM(+bias, GELU epilogue) ->
        GEMM(+bias, +residual; layer-scale folded into the weights); downsample = LN + 2x2/s2 patch gather + GEMM.
        images [B,3,768,768] -> [B*576, C_last]."""
        cfg, fz, dev = self.cfg, self.fz, self.dev
        B, _, Hi, Wi = images.shape
        g = Hi // 4
        cols = images.to(BF16).view(B, 3, g, 4, g, 4).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 48)
        a = torch.zeros(B * g * g, 64, device=dev, dtype=BF16)
        a[:, :48] = cols
        x = ops.gemm(a, fz["cnx.stem_w"], bias=fz["cnx.stem_b"])
        x, _, _ = ops.layernorm_fwd(x, fz["cnx.stem_ln_w"], fz["cnx.stem_ln_b"], cfg.cnx_eps, save_stats=False)
        Hc = Wc = g
        for i, dep in enumerate(cfg.cnx_depths):
            C = cfg.cnx_dims[i]
            if i > 0:
                Cp = cfg.cnx_dims[i - 1]
                x, _, _ = ops.layernorm_fwd(x, fz[f"cnx.{i}.ds_ln_w"], fz[f"cnx.{i}.ds_ln_b"], cfg.cnx_eps, save_stats=False)
                key = ("cnx_ds", B, Hc, Wc)
                if key not in self._static:                          # 2x2/s2 patch rows: (b, y2, x2, dy, dx) -> source pixel row
                    bb = torch.arange(B, device=dev).view(B, 1, 1, 1, 1)
                    y2 = torch.arange(Hc // 2, device=dev).view(1, -1, 1, 1, 1)
                    x2 = torch.arange(Wc // 2, device=dev).view(1, 1, -1, 1, 1)
                    dy = torch.arange(2, device=dev).view(1, 1, 1, 2, 1)
                    dx = torch.arange(2, device=dev).view(1, 1, 1, 1, 2)
                    self._static[key] = ((bb * Hc + 2 * y2 + dy) * Wc + 2 * x2 + dx).reshape(-1).to(torch.int32)
                rows = self._static[key]
                Hc, Wc = Hc // 2, Wc // 2
                patches = torch.empty(B * Hc * Wc * 4, Cp, device=dev, dtype=BF16)
                ops.gather_rows([x], torch.zeros(rows.numel(), device=dev, dtype=torch.int32), rows, Cp, patches)
                x = ops.gemm(patches.view(B * Hc * Wc, 4 * Cp), fz[f"cnx.{i}.ds_w"], bias=fz[f"cnx.{i}.ds_b"])
            for j in range(dep):
                o = f"cnx.{i}.{j}."
                y = ops.dwconv7x7_nhwc(x.view(B, Hc, Wc, C), fz[o + "dw_w"], fz[o + "dw_b"]).view(-1, C)
                y, _, _ = ops.layernorm_fwd(y, fz[o + "ln_w"], fz[o + "ln_b"], cfg.cnx_eps, save_stats=False)
                y = ops.gemm(y, fz[o + "w1"], bias=fz[o + "b1"], epi=ops.EPI_GELU)
                x = ops.gemm(y, fz[o + "w2"], bias=fz[o + "b2"], residual=x)
        return x                                                      # [B*Hc*Wc, C_last], rows already (b, y, x)

    def vit_forward(self, images):
        """Frozen vision tower -> [B*576, C] bf16 (no grad): CLIP-ViT hidden_states[select_layer][:, 1:], or the ConvNeXt trunk."""
        if getattr(self, "has_convnext", False):
            return self.convnext_forward(images)
        cfg, fz = self.cfg, self.fz
        B = images.shape[0]
        g = cfg.vit_image // cfg.vit_patch
        P, C, nh = cfg.vit_patch, cfg.vit_hidden, cfg.vit_heads
        N = g * g + 1
        # im2col (pure data movement; conv stride == kernel): [B*g*g, 3*P*P] zero-padded to the GEMM K multiple
        cols = images.to(BF16).view(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * P * P)
        kp = fz["vit.patch_w"].shape[1]
        a = torch.zeros(B * g * g, kp, device=self.dev, dtype=BF16)
        a[:, :3 * P * P] = cols
        h = torch.empty(B, N, C, device=self.dev, dtype=BF16)
        pkey = ("vit_pos", B)
        if pkey not in self._static:                                 # learned positions repeated per sample: the GEMM's residual operand
            self._static[pkey] = fz["vit.pos"].repeat(B, 1).contiguous()
        pe = ops.gemm(a, fz["vit.patch_w"], residual=self._static[pkey])          # ONE GEMM for the whole batch (+ positions in the epilogue)
        ops.copy2d_(h.view(B, N * C)[:, C:], pe.view(B, g * g * C))                  # patch rows behind each sample's CLS row
        h[:, 0] = fz["vit.cls_pos"]
        x, _, _ = ops.layernorm_fwd(h.view(B * N, C), fz["vit.pre_layrnorm.weight"], fz["vit.pre_layrnorm.bias"], cfg.vit_eps,
                                    save_stats=False)
        hd = C // nh
        for l in range(self.vit_layers_run()):
            o = f"vit.{l}."
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln1w"], fz[o + "ln1b"], cfg.vit_eps, save_stats=False)
            qkv = ops.gemm(y, fz[o + "wqkv"], bias=fz[o + "bqkv"]).view(B, N, 3 * C)
            att, _ = ops.attn_fwd(qkv[..., :C].view(B, N, nh, hd), qkv[..., C:2 * C].view(B, N, nh, hd),
                                  qkv[..., 2 * C:].view(B, N, nh, hd), causal=False)
            x = ops.gemm(att.view(B * N, C), fz[o + "wo"], bias=fz[o + "bo"], residual=x)
            y, _, _ = ops.layernorm_fwd(x, fz[o + "ln2w"], fz[o + "ln2b"], cfg.vit_eps, save_stats=False)
            y = ops.gemm(y, fz[o + "w1"], bias=fz[o + "b1"], epi=ops.EPI_QUICK_GELU)
            x = ops.gemm(y, fz[o + "w2"], bias=fz[o + "b2"], residual=x)
        x = x.view(B, N, C)
        if cfg.mm_vision_select_feature == "patch":
            x = x[:, 1:]
        return x.contiguous().view(-1, C)

    # ------------------------------------------------------------------------------------------ a11: DPT depth decoder
    def dpt_forward(self, feats):
        """DAv2_Head.forward + DPTHead.forward (da_v2_head.py:260-293, 316-321) and the min-max normalisation of
        base_ola_vlm.py:466-468, frozen / no grad.  feats: 4 x [B, 576, 1024] bf16 -> depth_pred [B, 336, 336] bf16."""
        fz = self.fz
        if "dpt.proj0.w" not in fz:
            raise RuntimeError("the DPT depth decoder weights (da_v2_head.*) were not loaded")
        B, P = feats[0].shape[0], int(round(feats[0].shape[1] ** 0.5))
        outs = []
        for i, x in enumerate(feats):
            x = ops.gemm(x.reshape(B * P * P, -1), fz[f"dpt.proj{i}.w"], bias=fz[f"dpt.proj{i}.b"]).view(B, P, P, -1)
            if i == 0:
                x = ops.conv_transpose_nhwc(x, fz["dpt.up0.w"], fz["dpt.up0.b"], 4)
            elif i == 1:
                x = ops.conv_transpose_nhwc(x, fz["dpt.up1.w"], fz["dpt.up1.b"], 2)
            elif i == 3:
                x = ops.conv3x3_nhwc(x, fz["dpt.down3.w"], bias=fz["dpt.down3.b"], stride=2)
            outs.append(x)
        rn = [ops.conv3x3_nhwc(outs[i], fz[f"dpt.rn{i}.w"]) for i in range(4)]

        def rcu(x, p):                                   # ResidualConvUnit: conv2(relu(conv1(relu(x)))) + x
            t = ops.conv3x3_nhwc(x, fz[p + "c1.w"], bias=fz[p + "c1.b"], relu_in=True, epi=ops.EPI_RELU)
            return ops.conv3x3_nhwc(t, fz[p + "c2.w"], bias=fz[p + "c2.b"], residual=x)

        def fusion(r, x0, x1=None, size=None):           # FeatureFusionBlock
            out = x0
            if x1 is not None:
                out = ops.add(out, rcu(x1, f"dpt.ref{r}.rcu1."))
            out = rcu(out, f"dpt.ref{r}.rcu2.")
            Ho, Wo = size if size is not None else (2 * out.shape[1], 2 * out.shape[2])
            out = ops.bilinear_nhwc(out, Ho, Wo)
            return ops.gemm(out.view(-1, out.shape[-1]), fz[f"dpt.ref{r}.out.w"], bias=fz[f"dpt.ref{r}.out.b"]).view(B, Ho, Wo, -1)

        path4 = fusion(4, rn[3], size=rn[2].shape[1:3])
        path3 = fusion(3, path4, rn[2], size=rn[1].shape[1:3])
        path2 = fusion(2, path3, rn[1], size=rn[0].shape[1:3])
        path1 = fusion(1, path2, rn[0])
        out = ops.conv3x3_nhwc(path1, fz["dpt.oc1.w"], bias=fz["dpt.oc1.b"])
        out = ops.bilinear_nhwc(out, 14 * P, 14 * P)
        out = ops.conv3x3_nhwc(out, fz["dpt.oc2a.w"], bias=fz["dpt.oc2a.b"], epi=ops.EPI_RELU)
        out = ops.gemm(out.view(-1, out.shape[-1]), fz["dpt.oc2b.w"], bias=fz["dpt.oc2b.b"], epi=ops.EPI_RELU)   # [pixels, 8]; col 0
        depth = out[:, 0].contiguous().view(B, 14 * P * 14 * P)       # relu(relu(x)) == relu(x): DAv2_Head's extra F.relu is a no-op
        return ops.minmax_norm(depth).view(B, 14 * P, 14 * P)

    # ------------------------------------------------------------------------------------------ linear helpers
    def _wgrad(self, x2d, dy2d, gview, accumulate=False):
        """gview[N,K] (fp32) (+)= dy^T @ x.  Aligned shapes go straight from the row-major activations through the TN kernel
        (no transposed copies); the rest via two zero-padded transposes + the NT GEMM."""
        M = x2d.shape[0]
        N, K = dy2d.shape[1], x2d.shape[1]
        g2 = gview.view(N, K)
        if (ops.gemm_tn_ok(N, K, M) and dy2d.stride(0) % 8 == 0 and x2d.stride(0) % 8 == 0 and dy2d.stride(1) == 1 and x2d.stride(1) == 1
                and (dy2d.data_ptr() | x2d.data_ptr() | g2.data_ptr()) % 16 == 0):
            ops.gemm_tn(dy2d, x2d, out=g2, accumulate=accumulate)
            return
        Mp = (M + 63) // 64 * 64
        # zero-padded transposed operands live in a small pool keyed by (role, rows, M): the pad columns are zeroed once, the transposes only
        # ever rewrite columns < M, and every use is ordered on the one stream (was: two fill kernels per call, ~230 launches per step)
        # one pool per STREAM (the heads' weight gradients run on the side stream, the projector's on the main stream: a shared buffer would be
        # rewritten by one stream while a GEMM of the other still reads it); when a pool overflows (ragged batches: M changes every step) the
        # stream is drained before its buffers are dropped, so no queued kernel can still read a freed block
        st = torch.cuda.current_stream()
        pool = self.__dict__.setdefault("_wgrad_pool", {}).setdefault(st.cuda_stream, {})

        def padded(role, rows):
            key = (role, rows, M)
            if key not in pool:
                if len(pool) >= 64:
                    st.synchronize()
                    pool.clear()
                pool[key] = torch.zeros(rows, Mp, device=self.dev, dtype=BF16)
            return pool[key]
        dyT, xT = padded(0, N), padded(1, K)
        ops.transpose(dy2d, out=dyT)
        ops.transpose(x2d, out=xT)
        if accumulate:
            tmp = ops.gemm(dyT, xT, out_f32=True)
            ops._lib.call("vp_colsum_finish", 1, N * K, ops._p(tmp), ops._p(g2), 1.0, 1, ops._stream())
        else:
            ops.gemm(dyT, xT, out=g2)

    def _wT(self, key, w):
        """[in, out] copy of a trainable weight for its dgrad GEMM, cached until the bf16 shadow changes (ParamStore.version)."""
        c = self.__dict__.setdefault("_wT_cache", {"v": -1, "t": {}})
        if c["v"] != self.ps.version:
            c["v"], c["t"] = self.ps.version, {}
        if key not in c["t"]:
            c["t"][key] = _tp(w)
        return c["t"][key]

    def _lin_bwd(self, x2d, dy2d, wname, bname=None, need_dx=True):
        ps = self.ps
        dx = None
        if need_dx:
            dx = ops.gemm(dy2d, self._wT(wname, ps.w(wname)))
        self._wgrad(x2d, dy2d, ps.g(wname))
        if bname is not None:
            ops.colsum(dy2d, out=ps.g(bname))
        return dx

    def _acc(self, gview, src_f32, scale=1.0):
        """gview += scale * src (fp32, flat)."""
        n = src_f32.numel()
        ops._lib.call("vp_colsum_finish", 1, n, ops._p(src_f32), ops._p(gview), scale, 1, ops._stream())

    # ------------------------------------------------------------------------------------------ splice plan (host)
    def image_groups(self, images):
        """`images` as the reference's prepare_inputs_labels_for_multimodal accepts it (ola_arch.py:262-275) -> (one [N, 3, H, W] tensor for the
        tower, group_sizes or None).  4-D tensor: one image per <image> token (None).  list of [3, H, W] / [n_j, 3, H, W] tensors or a 5-D
        [B, n, 3, H, W] tensor: entry j's n_j images are encoded together and their features FLATTENED to n_j * 576 rows that replace ONE
        <image> token (mm_patch_merge_type "flat", the default and the PT / IFT scripts' setting).  The "spatial" / "anyres" merges
        (ola_arch.py:276-308: image_newline, unpad) are not built and refused."""
        if torch.is_tensor(images) and images.dim() == 4:
            return images, None
        merge = getattr(self.cfg, "mm_patch_merge_type", "flat")
        if merge != "flat":
            raise NotImplementedError(f"mm_patch_merge_type={merge!r}: only the 'flat' merge of list / 5-D `images` is built "
                                      f"(ola_arch.py:272-275); 'spatial*' / anyres (ola_arch.py:276-308) is out of scope")
        if isinstance(images, (list, tuple)):
            ims = [x.unsqueeze(0) if x.dim() == 3 else x for x in images]                   # ola_arch.py:264-265
            if not ims or any(x.dim() != 4 for x in ims):
                raise ValueError("a list `images` must hold [3, H, W] or [n, 3, H, W] tensors (ola_arch.py:263-266)")
            if any(x.shape[1:] != ims[0].shape[1:] for x in ims):
                raise ValueError("the entries of a list `images` must share (3, H, W): torch.cat in ola_arch.py:266 raises otherwise")
            return torch.cat(ims, 0), [int(x.shape[0]) for x in ims]
        if torch.is_tensor(images) and images.dim() == 5:
            return images.reshape(-1, *images.shape[2:]), [int(images.shape[1])] * int(images.shape[0])
        raise ValueError(f"`images` must be a 4-D / 5-D tensor or a list of tensors (ola_arch.py:262), got {type(images).__name__}")

    def build_plan(self, input_ids, attention_mask, labels, group_sizes=None):
        """splice.host_plan (the index bookkeeping of prepare_inputs_labels_for_multimodal, ola_arch.py:256-444) + ONE pinned
        int32 buffer -> one async H2D copy of every table.  A fresh batch every step costs ~1 ms of host time, which runs under the
        previous step's GPU work; identical batches hit a small cache."""
        from . import splice
        cfg = self.cfg
        ids = np.ascontiguousarray(input_ids.detach().cpu().numpy().astype(np.int64, copy=False))
        am = None if attention_mask is None else attention_mask.detach().cpu().numpy().astype(bool, copy=False)
        lab = None if labels is None else labels.detach().cpu().numpy().astype(np.int64, copy=False)
        key = (ids.tobytes(), None if am is None else am.tobytes(), None if lab is None else lab.tobytes(), cfg.tokenizer_padding_side,
               None if group_sizes is None else tuple(group_sizes))
        if key in self._plan_cache:
            return self._plan_cache[key]
        hp = splice.host_plan(cfg, self.tasks, ids, am, lab, group_sizes)
        plan = {k: hp[k] for k in ("B", "S", "n_img", "n_feat", "n_valid", "lens_host", "n_tok_rows", "full", "side", "tok_cnt")}
        for k in ("labels", "attention_mask", "position_ids"):
            plan[k] = torch.from_numpy(hp[k])
        tabs = hp["tables"]                                           # name -> int32 array
        # the compacted lm_head rows are padded to whole 256-row GEMM tiles with rows that carry no label (gathered as zeros, label -100: zero
        # loss, zero d_logits, never scattered back) so that every chunk of the two vocabulary-wide GEMMs takes the aligned 4-wave kernel
        ce_labels = hp["ce_labels"]
        n_ce = int(tabs["ce_rows"].size)
        tabs = dict(tabs, ce_dst=tabs["ce_rows"], un_kind=np.zeros(tabs["un_rows"].size, np.int32), un_dst=tabs["un_rows"])
        if n_ce >= 512 and n_ce % 256:
            pad = 256 - n_ce % 256
            tabs = dict(tabs, ce_rows=np.concatenate([tabs["ce_rows"], np.zeros(pad, np.int32)]),
                        ce_kind=np.concatenate([tabs["ce_kind"], np.full(pad, -1, np.int32)]),
                        ce_dst=np.concatenate([tabs["ce_dst"], np.full(pad, -1, np.int32)]))
            ce_labels = np.concatenate([ce_labels, np.full(pad, IGNORE_INDEX, ce_labels.dtype)])
            n_ce += pad
        n_un = int(tabs["un_rows"].size)                              # same padding for the label-less rows' forward-only lm_head pass (keep_logits)
        if n_un >= 512 and n_un % 256:
            pad = 256 - n_un % 256
            tabs = dict(tabs, un_rows=np.concatenate([tabs["un_rows"], np.zeros(pad, np.int32)]),
                        un_kind=np.concatenate([tabs["un_kind"], np.full(pad, -1, np.int32)]),
                        un_dst=np.concatenate([tabs["un_dst"], np.full(pad, -1, np.int32)]))
            n_un += pad
        offs, tot = {}, 0
        for name, a in tabs.items():
            offs[name] = (tot, a.size)
            tot += (a.size + 3) // 4 * 4                              # 16-byte aligned slices
        host = torch.empty(max(tot, 4), dtype=torch.int32, pin_memory=True)
        hv = host.numpy()
        for name, a in tabs.items():
            o, n = offs[name]
            hv[o:o + n] = a.reshape(-1)
        devbuf = host.to(self.dev, non_blocking=True)
        M = plan["B"] * plan["S"]
        plan["n_ce"], plan["n_un"] = n_ce, n_un
        shift_h = torch.from_numpy(np.concatenate([hp["shift_labels"], ce_labels])).pin_memory()
        plan["_host_bufs"] = (host, shift_h)                          # keep the pinned sources alive until the async copies ran
        shift_d = shift_h.to(self.dev, non_blocking=True)
        plan["shift_labels"], plan["ce_labels"] = shift_d[:M], shift_d[M:]
        view = lambda name: devbuf[offs[name][0]:offs[name][0] + offs[name][1]]
        for name in ("kind", "row", "lens", "img_dst", "tok_src", "embed_idx", "ce_rows", "ce_inv", "ce_kind", "ce_inv_kind", "ce_dst", "un_rows",
                     "un_kind", "un_dst"):
            plan[name] = view(name)
        plan["present"] = (view("present_kind"), view("present")) if "present" in offs else None
        heads = hp["heads"]
        for task, h in heads.items():
            h["rows"] = view("rows:" + task)
            h["xin"] = dict(kind=view("xin_kind:" + task), row=view("xin_row:" + task),
                            mean_idx=view("mean_idx:" + task) if ("mean_idx:" + task) in offs else None,
                            lat_bwd=view("lat_bwd:" + task) if ("lat_bwd:" + task) in offs else None, lat_cnt=h.get("lat_cnt", 0),
                            lat_rep=view("lat_rep:" + task) if ("lat_rep:" + task) in offs else None)
        plan["heads"] = heads
        plan["inv"] = {l: view(f"inv:{l}") for l in self.tapped if f"inv:{l}" in offs}
        if len(self._plan_cache) > 8:
            self._plan_cache.clear()
        self._plan_cache[key] = plan
        return plan

    # ------------------------------------------------------------------------------------------ embed
    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            # lowest priority the device offers (VP_HEADS_PRIO overrides; gfx950 / ROCm 7: range (0, -1), no measurable difference between them)
            lo_pri = torch.cuda.Stream.priority_range()[0]
            self._side = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("VP_HEADS_PRIO", lo_pri)))
        return self._side

    def _embed(self, images, plan, images_resident=False):
        """CLIP tower -> mlp2x_gelu projector -> task-token rows -> splice gather.  Returns x [B*S,H] and the projector
        activations its backward needs.
        images_resident: the caller states that `images` was fully written before any work still pending on the current stream was enqueued (bench.py:
        synthetic batches resident in HBM; a dataloader that copies to the device on its own stream).  The frozen tower then runs on the side
        stream WITHOUT waiting for the current stream: it depends on nothing but the images and the frozen weights, so when the host runs ahead
        its ~160 small launches execute under the previous step's decoder backward (in its XCD tails and dispatch gaps) instead of at the head of
        this step with the chip to themselves.  Every step still runs its own tower pass; only its place in the GPU's schedule moves.
        VP_TOWER_STREAM=0: on the current stream, as before."""
        cfg, fz, ps, dev = self.cfg, self.fz, self.ps, self.dev
        H = cfg.hidden_size
        if images_resident and os.environ.get("VP_TOWER_STREAM", "1") != "0":
            main, side = torch.cuda.current_stream(), self._side_stream()
            after = os.environ.get("VP_TOWER_AFTER")                # dev aid (tools/nan_bisect_r06.sh): the tower may only start once the PREVIOUS
            ev_prev = getattr(self, "_phase_ev", {}).get(after)     # step's decoder forward ("fwd") / backward ("bwd") is done on the main stream
            if ev_prev is not None:
                side.wait_event(ev_prev)
            with torch.cuda.stream(side):
                feats = self.vit_forward(images)
                ev = torch.cuda.Event()
                ev.record(side)
            main.wait_event(ev)
            feats.record_stream(main)
        else:
            feats = self.vit_forward(images)                                       # [n_img*576, C]
        z1 = ops.gemm(feats, ps.w("model.mm_projector.0.weight"), bias=ps.w("model.mm_projector.0.bias"))
        a1 = ops.act_fwd(z1, ops.EPI_GELU)
        img = ops.gemm(a1, ps.w("model.mm_projector.2.weight"), bias=ps.w("model.mm_projector.2.bias"))
        # task-token rows (a4).  PT stage (ola_arch.py:224-254) and the IFT stage's "expand_emb": depth / seg = group means of the (576, H)
        # parameter, gen = its raw rows; IFT stage "emb" (llava_arch.py:259-260): every parameter row as it is
        tok_rows = None
        if plan["n_tok_rows"] > 0:
            tok_rows = torch.empty(plan["n_tok_rows"], H, device=dev, dtype=BF16)
            off = 0
            for task, nr, pooled in task_token_rows(cfg):
                src = ps.w(f"model.special_{task}_tokens")
                if not pooled:
                    ops.copy2d_(tok_rows[off:off + nr], src)
                else:
                    grp = src.shape[0] // nr
                    ops.gather_sum_rows(src, self._arange(src.shape[0]), grp, 1.0 / grp, tok_rows[off:off + nr])
                off += nr
        x = torch.empty(plan["B"] * plan["S"], H, device=dev, dtype=BF16)
        srcs = [fz["embed"], img] + ([tok_rows] if tok_rows is not None else [])
        ops.gather_rows(srcs, plan["kind"], plan["row"], H, x)
        return x, feats, z1, a1, img

    def present(self, x2d, plan):
        """[B*S, C] rows in the kernels' left-aligned layout -> [B, S, C] as the reference lays them out (identity for right
        padding; right-aligned with zero pad rows for tokenizer_padding_side == "left")."""
        B, S = plan["B"], plan["S"]
        if plan["present"] is None:
            return x2d.view(B, S, -1)
        out = torch.empty_like(x2d)
        ops.gather_rows([x2d], plan["present"][0], plan["present"][1], x2d.shape[-1], out)
        return out.view(B, S, -1)

    def splice_forward(self, input_ids, attention_mask, labels, images):
        """prepare_inputs_labels_for_multimodal's tensor outputs (ola_arch.py:256-444): (inputs_embeds [B,S,H], plan)."""
        images, gs = self.image_groups(images)
        plan = self.build_plan(input_ids, attention_mask, labels, gs)
        x, *_ = self._embed(images.to(self.dev), plan)
        return self.present(x, plan), plan

    # ------------------------------------------------------------------------------------------ the step
    def train_step(self, batch, compute_grads=True):
        """One fused forward+backward.  Returns dict(loss, text_loss, per-task losses, layer_losses, ...) of device
        scalars (fp32 tensors); gradients of the trainable set are in self.ps.grad (zeroed first).
        Stages (each its own method): _embed (a1..a5) -> _decoder_fwd (a6) -> _ntp (a7) -> _heads (a8..a14) -> _decoder_bwd -> _splice_bwd."""
        ps = self.ps
        images, gs = self.image_groups(batch["images"])
        gs = batch.get("image_group_sizes", gs)                   # (a caller that already flattened a list / 5-D `images`: the model mirrors)
        plan = self.build_plan(batch["input_ids"], batch.get("attention_mask"), batch.get("labels"), gs)
        if compute_grads:
            if self.train_llm:
                # every decoder / lm_head / norm gradient is overwritten by its wgrad GEMM below: clear only what accumulates
                # (heads, projector, task tokens in front of the LLM block, and the scatter-added embedding table at its end)
                ops.zero_(ps.grad[:ps.index["lm_head.weight"][0]])
                ops.zero_(ps.g("model.embed_tokens.weight"))
            else:
                ps.zero_grad()
        out = {"plan": plan}

        # ---- vision tower + projector + splice (a1..a5)
        x, feats, z1, a1, img = self._embed(images, plan, images_resident=bool(batch.get("images_resident", False)))
        out["image_features"] = img
        out["inputs_embeds"] = self.present(x, plan)

        dec = self._decoder_fwd(x, plan, compute_grads)
        if os.environ.get("VP_TOWER_AFTER"):
            self.__dict__.setdefault("_phase_ev", {})["fwd"] = torch.cuda.current_stream().record_event()
        out["hidden"] = self.present(dec["hidden"], plan)
        out["layer_states"] = {l: self.present(t, plan) for l, t in dec["states"].items()}      # the tapped states the heads read (views when right-padded)
        if dec["hidden_states"] is not None:             # (embeddings, layer 1 .. L-1 outputs, norm(layer L output)): ola_llama.py:113,181
            out["hidden_states"] = (out["inputs_embeds"],) + tuple(self.present(t, plan) for t in dec["hidden_states"][1:-1]) + (out["hidden"],)
        # The distillation heads (a8..a14: ~300 small launches, few tiles each) depend only on the tapped layer states, and nothing needs their
        # result before the decoder backward reaches the topmost tapped layer.  They run on a SIDE STREAM, forked here and joined there
        # (_join_heads), so their workgroups fill the CUs the persistent GEMMs of lm_head / the upper layers' backward leave idle in their
        # XCD tails and dispatch gaps instead of owning the chip for ~13 ms.  Same kernels, same inputs, no atomics: results are bit-identical
        # to the serial order (VP_HEADS_STREAM=0).  The reference runs them serially inside forward (base_ola_vlm.py:445-534).
        fork = compute_grads and len(self.tasks) > 0 and os.environ.get("VP_HEADS_STREAM", "1") != "0"
        self._heads_join = None
        if fork:
            main = torch.cuda.current_stream()
            side = self._side_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                task_loss, d_state = self._heads(dec["states"], plan, batch, compute_grads, out)
                if self.world > 1:
                    self._reducer().start_early()                # heads + logit scales: overlap with the decoder backward
                self._heads_join = torch.cuda.Event()
                self._heads_join.record(side)
            # Everything _heads allocated on the side stream and handed on (layer-state gradients, per-task / per-layer losses, embeddings, depth
            # maps) is READ on the main stream (total(), the caller).  Tell the allocator: otherwise a caller that drops `out` early returns those
            # blocks to the side stream's pool while main-stream reads are still queued, and the next step's side-stream tower may reuse them
            # first.
            def _rs(v):
                if torch.is_tensor(v):
                    if v.is_cuda:
                        v.record_stream(main)
                elif isinstance(v, dict):
                    for x in v.values():
                        _rs(x)
                elif isinstance(v, (list, tuple)):
                    for x in v:
                        _rs(x)
            _rs(d_state)
            _rs(task_loss)
            for k in ("layer_losses", "embs", "depth_preds", "depth_feats"):
                _rs(out.get(k))
            text_loss, d_hidden = self._ntp(dec["hidden"], plan, compute_grads, out)
        else:
            text_loss, d_hidden = self._ntp(dec["hidden"], plan, compute_grads, out)
            task_loss, d_state = self._heads(dec["states"], plan, batch, compute_grads, out)
            if compute_grads and self.world > 1:
                self._reducer().start_early()                    # heads + logit scales: overlap with the decoder backward

        def total():
            loss = text_loss.clone()
            for task in ("seg", "depth", "gen"):                             # sum order: ola_llama.py:143-144
                if task in task_loss:
                    loss = loss + task_loss[task]
                    out[f"{task}_loss"] = task_loss[task]
            out["loss"] = loss
        if not compute_grads:
            total()
            return out
        dx = self._decoder_bwd(d_hidden, dec, d_state, plan)
        if os.environ.get("VP_TOWER_AFTER"):
            self.__dict__.setdefault("_phase_ev", {})["bwd"] = torch.cuda.current_stream().record_event()
        self._join_heads()
        total()
        out["d_inputs_embeds"] = self.present(dx, plan)
        self._splice_bwd(dx, plan, feats, z1, a1)
        return out

    def _join_heads(self):
        """The current stream waits (on the GPU; the host does not block) for the side stream's heads: called in front of the first use of their
        results (d_state in the decoder backward) and once more at its end."""
        if getattr(self, "_heads_join", None) is not None:
            torch.cuda.current_stream().wait_event(self._heads_join)
            self._heads_join = None

    def _attn_window(self):
        """kernels keep keys with q - key < window; transformers 4.41.1 (the reference's pin) keeps q - key <= sliding_window"""
        cfg = self.cfg
        return (int(cfg.sliding_window) + int(bool(getattr(cfg, "sliding_window_inclusive", False)))) if cfg.sliding_window else 0

    def _qkv_views(self, t, B, S):
        cfg = self.cfg
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        t3 = t.view(B, S, -1)
        return (t3[..., :nh * hd].view(B, S, nh, hd), t3[..., nh * hd:(nh + nkv) * hd].view(B, S, nkv, hd),
                t3[..., (nh + nkv) * hd:].view(B, S, nkv, hd))

    def _decoder_fwd(self, x, plan, compute_grads):
        """a6: the decoder stack (ola_llama.py:105-119 -> HF LlamaModel / Phi3Model).  Returns the final pre-norm state, the normed hidden
        state, what the backward pass needs per layer, and the tapped layer states (layer_states[i] = output of layer i + 1, the last one
        post-norm: ola_llama.py:117-119)."""
        cfg, fz = self.cfg, self.fz
        B, S = plan["B"], plan["S"]
        M = B * S
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        cos_t, sin_t = self.rope(S)
        kv_len = None if plan["full"] else plan["lens"]
        window = self._attn_window()
        L = cfg.num_hidden_layers
        saved, states = [], {}
        il = self.gu_interleaved
        fuse = il and ops.swiglu_fusable(M, 2 * cfg.intermediate_size, cfg.hidden_size) and \
            ops.swiglu_fusable(M, cfg.intermediate_size, cfg.hidden_size)
        fuse_rope = ops.gemm_rope_ok(M, (nh + 2 * nkv) * hd, cfg.hidden_size, hd) and cos_t.shape[-1] == 64
        fold = getattr(self, "fold_norm", False) and not self.train_llm and fuse and (fuse_rope or hd == 96) and \
            ops.fold_norm_ok(M, cfg.hidden_size, cfg.intermediate_size, hd, (nh + 2 * nkv) * hd)
        self.last_fold = bool(fold)                      # (which set of rounding points the last step took: DESIGN section 2, deviation v)
        H, eps = cfg.hidden_size, cfg.rms_norm_eps
        rstd1 = None
        hs = [] if self.keep_states else None            # HF all_hidden_states: the input of every layer, then the post-norm final state
        for l in range(L):
            o = f"dec.{l}."
            if hs is not None:
                hs.append(x)
            if fold:
                # no normalised copy of the stream: 1/rms rides in the consuming GEMM's epilogue (gamma is in its weight), the statistics of the next
                # norm come out of the residual GEMM that writes the stream (layer 0's from one pass over the spliced embeddings)
                if rstd1 is None:
                    _, rstd1 = ops.rmsnorm_fwd(x, fz["ones_h"], eps)
                if fuse_rope:
                    qkv = ops.gemm_rope(x, fz[o + "wqkv"], S, (nh + nkv) * hd, cos_t, sin_t, row_scale=rstd1)
                else:                                       # D = 96: row-scaled GEMM (rope_cols = 0), then the rotation kernel
                    qkv = ops.gemm_rope(x, fz[o + "wqkv"], S, 0, cos_t, sin_t, row_scale=rstd1)
                    ops.rope_(qkv, M, S, nh + nkv, hd, cos_t, sin_t)
                q4, k4, v4 = self._qkv_views(qkv, B, S)
                att, lse = ops.attn_fwd(q4, k4, v4, causal=True, window=window, kv_len=kv_len)
                h1, part = ops.gemm_sumsq(att.view(M, nh * hd), fz[o + "wo"], x)
                rstd2 = ops.rstd_from_sumsq(part, H, eps)
                gu, act = ops.gemm_swiglu_fwd(h1, fz[o + "wgu"], row_scale=rstd2)
                if l == L - 1:
                    xo = ops.gemm(act, fz[o + "wd"], residual=h1)
                    rnext = None
                else:
                    xo, part = ops.gemm_sumsq(act, fz[o + "wd"], h1)
                    rnext = ops.rstd_from_sumsq(part, H, eps)
                del part
                if compute_grads:
                    saved.append((x, rstd1, qkv, att, lse, h1, rstd2, gu))
                if l in self.tapped and l != L - 1:
                    states[l] = xo
                x, rstd1 = xo, rnext
                continue
            xn, rstd1 = ops.rmsnorm_fwd(x, fz[o + "ln1"], cfg.rms_norm_eps)
            if fuse_rope:                                       # RoPE of q and k in the QKV GEMM's epilogue (bit-identical to gemm + rope_)
                qkv = ops.gemm_rope(xn, fz[o + "wqkv"], S, (nh + nkv) * hd, cos_t, sin_t)
            else:
                qkv = ops.gemm(xn, fz[o + "wqkv"])
                ops.rope_(qkv, M, S, nh + nkv, hd, cos_t, sin_t)
            q4, k4, v4 = self._qkv_views(qkv, B, S)
            att, lse = ops.attn_fwd(q4, k4, v4, causal=True, window=window, kv_len=kv_len)
            h1 = ops.gemm(att.view(M, nh * hd), fz[o + "wo"], residual=x)
            hn, rstd2 = ops.rmsnorm_fwd(h1, fz[o + "ln2"], cfg.rms_norm_eps)
            if fuse:
                gu, act = ops.gemm_swiglu_fwd(hn, fz[o + "wgu"])
            else:
                gu = ops.gemm(hn, fz[o + "wgu"])
                act = ops.swiglu_fwd(gu, interleaved=il)
            xo = ops.gemm(act, fz[o + "wd"], residual=h1)
            if compute_grads:
                saved.append((x, rstd1, qkv, att, lse, h1, rstd2, gu))
            if l in self.tapped and l != L - 1:
                states[l] = xo
            x = xo
        hidden, rstd_f = ops.rmsnorm_fwd(x, fz["norm"], cfg.rms_norm_eps)
        if (L - 1) in self.tapped:
            states[L - 1] = hidden              # layer_states[-1] is the post-norm state (ola_llama.py:117-119)
        if hs is not None:
            hs.append(hidden)
        return dict(x=x, hidden=hidden, rstd_f=rstd_f, saved=saved, states=states, hidden_states=hs)

    def _lm_chunk(self, Mc, H):
        """Rows per lm_head chunk: whole 256-row tiles, at most lm_chunk_rows (the bf16 logits of a chunk are the step's largest transient:
        rows x V x 2 bytes), chosen so that the d_hidden GEMM — K = V is so long that one 256 x 256 tile takes milliseconds, and it has only
        rows / 256 x H / 256 tiles — wastes the fewest persistent-grid rounds: 11 264 labelled rows as 2 x 5632 = 2 x (352 tiles = 1.4 rounds of
        256 CUs -> 2) = 4 rounds, as one chunk = 704 tiles = 3 rounds (measured in the step: 9.7 -> 8.1 ms for that GEMM)."""
        cus = 256
        tn = max(1, (H + 255) // 256)
        best = None
        n0 = max(1, (Mc + self.lm_chunk_rows - 1) // self.lm_chunk_rows)
        for n in range(n0, n0 + 4):
            R = ((Mc + n - 1) // n + 255) // 256 * 256
            rounds, r0 = 0, 0
            while r0 < Mc:
                rows = min(R, Mc - r0)
                rounds += -(-(((rows + 255) // 256) * tn) // cus)
                r0 += R
            if best is None or rounds < best[0]:
                best = (rounds, R)
        return best[1]

    def _ntp(self, hidden, plan, compute_grads, out):
        """a7: lm_head + NTP loss (ola_llama.py:121-136), row-chunked; dlogits -> d_hidden in the same sweep.  Only the rows that carry a
        label go through the two vocabulary-wide GEMMs and the cross-entropy (the others have zero loss and zero d_logits): their hidden rows
        are compacted by one row gather, and d_hidden is scattered back with zeros elsewhere.  When the caller wants the reference's `logits`
        (keep_logits; ola_llama.py:121-122: `lm_head(hidden_states).float()` of EVERY row) each chunk's bf16 logits are widened straight into
        their rows of ONE fp32 [B, S, V] tensor before the cross-entropy overwrites them, and the label-less rows take a forward-only lm_head
        pass (no CE, no d_hidden GEMM) into the same tensor.  Returns (text_loss, d_hidden or None)."""
        fz, ps, dev = self.fz, self.ps, self.dev
        H = self.cfg.hidden_size
        M = plan["B"] * plan["S"]
        n_valid = plan["n_valid"]
        gscale = 1.0 / n_valid if n_valid > 0 else float("nan")
        keep_logits = self.keep_logits
        if keep_logits == "lazy":                                 # the caller may never read them: hand out the recipe (same bits: logits_of)
            out["logits_fn"] = lambda h=hidden, pl=plan: self.logits_of(h, pl)
            keep_logits = False
        direct = keep_logits and plan["present"] is None          # (ragged LEFT padding re-lays the rows for presentation: old cat + gather path)
        compact = 0 < n_valid < M and (direct or not keep_logits) and not self.lm_head_all_rows
        if compact:
            Mc = plan["n_ce"]                                    # n_valid rounded up to whole 256-row tiles (pad rows: zeros, label -100)
            h_ce = torch.empty(Mc, H, device=dev, dtype=BF16)
            ops.gather_rows([hidden], plan["ce_kind"], plan["ce_rows"], H, h_ce)
            lab_ce = plan["ce_labels"]
        else:
            h_ce, lab_ce, Mc = hidden, plan["shift_labels"], M
        d_hce = torch.empty(Mc, H, device=dev, dtype=BF16) if compute_grads else None
        row_loss = torch.empty(Mc, device=dev, dtype=F32)
        logits_keep = [] if (keep_logits and not direct) else None
        logits_f32 = torch.empty(M, fz["lm_head"].shape[0], device=dev, dtype=F32) if direct else None
        R = self._lm_chunk(Mc, H)
        for r0 in range(0, Mc, R):
            r1 = min(Mc, r0 + R)
            lg = ops.gemm(h_ce[r0:r1], fz["lm_head"])
            if logits_keep is not None:
                logits_keep.append(lg.clone())
            if direct:
                ops.scatter_rows_to_f32(lg, plan["ce_dst"][r0:r1], logits_f32) if compact else ops.scatter_rows_to_f32(lg, None, logits_f32[r0:r1])
            ops.ce_fwd_bwd(lg, lab_ce[r0:r1], gscale, write_grad=compute_grads, out=row_loss[r0:r1])
            if compute_grads:
                ops.gemm(lg, fz["lm_head_T"], out=d_hce[r0:r1])
                if self.train_llm:                             # lm_head.weight.grad (+)= dlogits^T h, chunk by chunk
                    self._wgrad(h_ce[r0:r1], lg, ps.g("lm_head.weight"), accumulate=r0 > 0)
        d_hidden = d_hce
        if compact and compute_grads:
            d_hidden = torch.empty(M, H, device=dev, dtype=BF16)
            ops.gather_rows([d_hce], plan["ce_inv_kind"], plan["ce_inv"], H, d_hidden)        # rows without a label: zeros
        text_loss = ops.sum_f32(row_loss, gscale)
        out["text_loss"] = text_loss
        if direct and compact and plan["n_un"] > 0:           # label-less rows: lm_head forward only
            Mu = plan["n_un"]
            h_un = torch.empty(Mu, H, device=dev, dtype=BF16)
            ops.gather_rows([hidden], plan["un_kind"], plan["un_rows"], H, h_un)
            for r0 in range(0, Mu, self.lm_chunk_rows):
                r1 = min(Mu, r0 + self.lm_chunk_rows)
                ops.scatter_rows_to_f32(ops.gemm(h_un[r0:r1], fz["lm_head"]), plan["un_dst"][r0:r1], logits_f32)
        if direct:
            out["logits"] = logits_f32.view(plan["B"], plan["S"], -1)
        elif logits_keep is not None:
            out["logits"] = self.present(torch.cat(logits_keep, 0), plan).float()
        return text_loss, d_hidden

    def logits_of(self, hidden, plan):
        """fp32 `logits` [B, S, V] of every row from the final (post-norm) hidden state [B*S, H] exactly as the step itself would return them
        (ola_llama.py:121-122: bf16 lm_head, then `.float()`); bit-identical to the keep_logits=True output (each row's dot products do not
        depend on which rows share its GEMM: `test_labelled_row_compaction_is_exact`)."""
        fz = self.fz
        M, V = hidden.shape[0], fz["lm_head"].shape[0]
        if plan["present"] is not None:
            lg = [ops.gemm(hidden[r0:min(M, r0 + self.lm_chunk_rows)], fz["lm_head"]) for r0 in range(0, M, self.lm_chunk_rows)]
            return self.present(torch.cat(lg, 0), plan).float()
        lf = torch.empty(M, V, device=self.dev, dtype=F32)
        for r0 in range(0, M, self.lm_chunk_rows):
            r1 = min(M, r0 + self.lm_chunk_rows)
            ops.scatter_rows_to_f32(ops.gemm(hidden[r0:r1], fz["lm_head"]), None, lf[r0:r1])
        return lf.view(plan["B"], plan["S"], V)

    def _heads(self, states, plan, batch, compute_grads, out):
        """a8..a14: every distillation head on its layer state — all forwards, ONE loss launch each way for all of them
        (vp_emb_loss_{fwd,bwd}_multi; the reference calls _emb_loss head by head: base_ola_vlm.py:445-534), the heads' backward passes, and the
        scatter of their input gradients back to full-length layer-state gradients.  Returns (task_loss {task: scalar}, d_state {layer: [M, H]}).
        Peak memory: every head's saved activations stay live until its own backward ran, i.e. all heads' at once (the price of the single
        loss launch); bench.py reports the step's peak allocation (`peak_mem_gb`), DESIGN.md section 3 has the measured numbers."""
        cfg, dev = self.cfg, self.dev
        H = cfg.hidden_size
        B, S = plan["B"], plan["S"]
        M = B * S
        d_state, task_loss = {}, {}
        out["layer_losses"] = {}
        out["embs"] = {}
        dx_parts = {l: [] for l in self.tapped}
        if not (len(self.tasks) > 0 and S > cfg.num_sys_tokens):
            return task_loss, d_state
        targets = self._prepare_targets(batch, B)
        ctxs = [self._head_fwd(task, i, idx, states[idx], plan, batch, targets, compute_grads) for task, i, idx in self.tasks]
        live = [c for c in ctxs if c["tg"] is not None]
        for c0 in range(0, len(live), 8):
            grp = live[c0:c0 + 8]
            outs = ops.emb_loss_fwd_multi([c["pred2"] for c in grp], [c["tg"][0] for c in grp], [c["tg"][1] for c in grp],
                                          [c["scale"] for c in grp], [cfg.contrastive_loss_weight] * len(grp), rank=self.rank)
            for c, (loss3, coef) in zip(grp, outs):
                c["res"]["loss3"], c["coef"] = loss3, coef
            if compute_grads:
                dps = ops.emb_loss_bwd_multi([c["pred2"] for c in grp], [c["tg"][0] for c in grp], [c["coef"] for c in grp],
                                             [c["w_t"] for c in grp], rank=self.rank)
                for c, dp in zip(grp, dps):
                    c["dpred"] = dp
        for c in ctxs:
            task, idx, res = c["task"], c["idx"], c["res"]
            if compute_grads and c["tg"] is not None:
                self._head_bwd(c)
            if res["loss3"] is not None:
                out["layer_losses"][(task, idx)] = res["loss3"]
                w_t = getattr(cfg, TASK_SPEC[task][0])[TASK_SPEC[task][2]]
                task_loss[task] = res["loss3"][0:1] * w_t if task not in task_loss else task_loss[task] + res["loss3"][0:1] * w_t
            out["embs"].setdefault(task, []).append(res["emb"])
            if "depth_pred" in res:
                out.setdefault("depth_preds", []).append(res["depth_pred"])
                out.setdefault("depth_feats", []).append(res["depth_feats"])
            if compute_grads and res["dx"] is not None:
                dx_parts[idx].append((task, res["dx"]))
        if not compute_grads:
            return task_loss, d_state
        # ---- scatter head input grads back to their layer states
        for l, parts in dx_parts.items():
            if not parts:
                continue
            tot = sum(p.shape[0] for _, p in parts)
            cat = torch.empty(tot, H, device=dev, dtype=BF16)
            off = 0
            for _, p in parts:
                ops.copy2d_(cat[off:off + p.shape[0]], p)
                off += p.shape[0]
            if len(parts) == len([1 for _, _, idx in self.tasks if idx == l]) and l in plan["inv"]:
                inv_t = plan["inv"][l]                   # built on the host with the plan (no D2H sync in the step)
            else:                                        # some head of this layer had no target: build the table for the heads that ran
                ikey = ("inv", l, tuple(t for t, _ in parts))
                if ikey not in plan:
                    inv = np.full((M, len(parts)), -1, np.int32)
                    off = 0
                    for j, (task, p) in enumerate(parts):
                        rows = plan["heads"][task]["rows_host"]
                        inv[rows, j] = off + np.arange(p.shape[0], dtype=np.int32)
                        off += p.shape[0]
                    plan[ikey] = torch.from_numpy(inv.reshape(-1)).to(dev)
                inv_t = plan[ikey]
            ds = torch.empty(M, H, device=dev, dtype=BF16)
            ops.gather_sum_rows(cat, inv_t, len(parts), 1.0, ds)
            d_state[l] = ds
        return task_loss, d_state

    def _decoder_bwd(self, d_hidden, dec, d_state, plan):
        """Backward of the decoder stack: dgrad only when the LLM is frozen (PT stage, ola_vlm_train.py:1127-1131), + weight gradients and
        per-layer gradient buckets when cfg.train_llm (IFT stage).  Returns d inputs_embeds [M, H]."""
        cfg, fz, ps = self.cfg, self.fz, self.ps
        B, S = plan["B"], plan["S"]
        M = B * S
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        L = cfg.num_hidden_layers
        cos_t, sin_t = self.rope(S)
        kv_len = None if plan["full"] else plan["lens"]
        window = self._attn_window()
        x, rstd_f, saved = dec["x"], dec["rstd_f"], dec["saved"]
        if (L - 1) in d_state:
            self._join_heads()
            ops.add(d_hidden, d_state[L - 1], out=d_hidden)
        train_llm = self.train_llm
        if train_llm:
            ops.rmsnorm_bwd_w(d_hidden, x, rstd_f, out=ps.g("model.norm.weight"))
            i0 = ps.index["lm_head.weight"][0]
            i1 = ps.index["model.norm.weight"][0] + 64 * ((ps.index["model.norm.weight"][1] + 63) // 64)
            self._reduce_range(i0, i1 - i0)                   # lm_head + final norm gradients are final: reduce under the backward
        dx = ops.rmsnorm_bwd(d_hidden, x, fz["norm"], rstd_f)
        del d_hidden
        for l in range(L - 1, -1, -1):
            o = f"dec.{l}."
            x_in, rstd1, qkv, att, lse, h1, rstd2, gu = saved[l]
            if l in d_state and l != L - 1:
                self._join_heads()
                ops.add(dx, d_state[l], out=dx)
            pl = f"model.layers.{l}."
            if self.gu_interleaved and ops.swiglu_fusable(M, cfg.intermediate_size, cfg.hidden_size) and gu.is_contiguous():
                d_gu = ops.gemm_swiglu_bwd(dx, fz[o + "wd_T"], gu)
            else:
                d_act = ops.gemm(dx, fz[o + "wd_T"])
                d_gu = ops.swiglu_bwd(d_act, gu, interleaved=self.gu_interleaved)
                del d_act
            if train_llm:                                       # weight gradients; cheap activations (act, hn) are recomputed
                self._wgrad(ops.swiglu_fwd(gu, interleaved=False), dx, ps.g(pl + "mlp.down_proj.weight"))
                hn, _ = ops.rmsnorm_fwd(h1, fz[o + "ln2"], cfg.rms_norm_eps, save_rstd=False)
                self._wgrad(hn, d_gu, self._gfused(o + "wgu"))
                del hn
            d_hn = ops.gemm(d_gu, fz[o + "wgu_T"])
            del d_gu
            if train_llm:
                ops.rmsnorm_bwd_w(d_hn, h1, rstd2, out=ps.g(pl + "post_attention_layernorm.weight"))
            d_h1 = ops.rmsnorm_bwd(d_hn, h1, fz[o + "ln2"], rstd2, dres=dx)
            if train_llm:
                self._wgrad(att.view(M, nh * hd), d_h1, ps.g(pl + "self_attn.o_proj.weight"))
            d_att = ops.gemm(d_h1, fz[o + "wo_T"])
            dqkv = torch.empty_like(qkv)
            q4, k4, v4 = self._qkv_views(qkv, B, S)
            dq4, dk4, dv4 = self._qkv_views(dqkv, B, S)
            if hd in (96, 128) and not os.environ.get("VP_NO_FUSED_ROPE"):     # RoPE^T of dq / dk fused into the attention-backward stores
                ops.attn_bwd(q4, k4, v4, att, lse, d_att.view(B, S, nh, hd), causal=True, window=window, kv_len=kv_len,
                             dq=dq4, dk=dk4, dv=dv4, rope=(cos_t, sin_t))
            else:
                ops.attn_bwd(q4, k4, v4, att, lse, d_att.view(B, S, nh, hd), causal=True, window=window, kv_len=kv_len,
                             dq=dq4, dk=dk4, dv=dv4)
                ops.rope_(dqkv, M, S, nh + nkv, hd, cos_t, sin_t, inverse=True)
            d_xn = ops.gemm(dqkv, fz[o + "wqkv_T"])
            if train_llm:
                xn, _ = ops.rmsnorm_fwd(x_in, fz[o + "ln1"], cfg.rms_norm_eps, save_rstd=False)
                self._wgrad(xn, dqkv, self._gfused(o + "wqkv"))
                del xn
                ops.rmsnorm_bwd_w(d_xn, x_in, rstd1, out=ps.g(pl + "input_layernorm.weight"))
                self._reduce_range(*self._llm_ranges[o])        # this layer's gradients are final
            dx = ops.rmsnorm_bwd(d_xn, x_in, fz[o + "ln1"], rstd1, dres=d_h1)
            saved[l] = None
        if train_llm:                                           # embed_tokens.weight.grad: scatter-add of the text rows
            ops.scatter_add_rows_(ps.g("model.embed_tokens.weight"), dx, plan["embed_idx"])
        return dx

    def _splice_bwd(self, dx, plan, feats, z1, a1):
        """Backward of the splice and the projector: image rows -> mm_projector (a3), task-token rows -> special-token parameters (a4)."""
        cfg, ps, dev = self.cfg, self.ps, self.dev
        H = cfg.hidden_size
        d_img = torch.empty(plan["n_feat"], H, device=dev, dtype=BF16)
        ops.gather_sum_rows(dx, plan["img_dst"], 1, 1.0, d_img)
        if plan["n_tok_rows"] > 0:
            d_tok = torch.empty(plan["n_tok_rows"], H, device=dev, dtype=F32)
            ops.gather_sum_rows(dx, plan["tok_src"], plan["tok_cnt"], 1.0, d_tok)
            off = 0
            for task, nr, pooled in task_token_rows(cfg):
                g = ps.g(f"model.special_{task}_tokens")
                if not pooled:                                   # raw rows: the parameter's gradient is the rows' own
                    self._acc(g, d_tok[off:off + nr])
                else:                                            # transpose of the group mean: every row of a group gets 1/grp of its pooled row's
                    grp = g.shape[0] // nr
                    key = ("tok_bwd", g.shape[0], grp, off)
                    if key not in self._static:
                        self._static[key] = (torch.arange(g.shape[0], device=dev, dtype=torch.int32) // grp + off).to(torch.int32)
                    ops.gather_sum_rows(d_tok, self._static[key], 1, 1.0 / grp, g, accumulate=True)
                off += nr
        d_a1 = self._lin_bwd(a1, d_img, "model.mm_projector.2.weight", "model.mm_projector.2.bias", need_dx=True)
        d_z1 = ops.act_bwd(d_a1, z1, ops.EPI_GELU)
        self._lin_bwd(feats, d_z1, "model.mm_projector.0.weight", "model.mm_projector.0.bias", need_dx=False)

    # ------------------------------------------------------------------------------------------ targets
    def _prepare_targets(self, batch, B=None):
        """Frozen-teacher features are inputs.  Flatten to [B, D] in the PREDICTION's memory order
        (seg targets (B,C,24,24) are re-laid out to (B,576,C) once so the loss kernel streams both linearly), then
        all-gather across DP ranks for the contrastive negatives (ola_utils.py:96-106), once per task per step."""
        out = {}
        for task in sorted({t for t, _, _ in self.tasks}):        # SORTED: every rank must issue the all-gathers in the same order
            tg = batch.get(f"{task}_target")
            if tg is None:
                out[task] = None
                continue
            tg = tg.to(device=self.dev, dtype=BF16)
            Bn = tg.shape[0]
            if task == "seg" and tg.dim() == 4:
                Cc = tg.shape[1]
                t3 = tg.reshape(Bn, Cc, -1)
                if (Cc * t3.shape[2]) % 8 == 0 and t3.is_contiguous():
                    flat = ops.transpose_batched(t3)                  # ONE launch (round 5: one transpose per sample)
                else:
                    flat = torch.empty(Bn, t3.shape[2], Cc, device=self.dev, dtype=BF16)
                    t3 = t3.contiguous()
                    for b in range(Bn):
                        ops.transpose(t3[b], out=flat[b])
                flat = flat.view(Bn, -1)
            else:
                flat = tg.reshape(Bn, -1).contiguous()
            from .parallel import all_gather_rows
            mask = batch.get(f"{task}_mask")
            mask = torch.ones(Bn, device=self.dev, dtype=F32) if mask is None else mask.to(device=self.dev, dtype=F32).reshape(-1)
            if B is not None and Bn != B:
                # _emb_loss's batch-repeat branch (base_ola_vlm.py:292-299): fewer target rows than predictions -> targets.repeat(r, 1, 1) and
                # mask.repeat(r, 1, 1) with r = B // Bn.  Like the reference this only works for rank-3 targets (its 3-argument repeat raises on
                # the (B, C, 24, 24) seg targets) and when B is a multiple of Bn (otherwise its smooth_l1_loss raises on the shapes)
                if tg.dim() != 3 or Bn == 0 or B % Bn != 0 or B < Bn:
                    raise ValueError(f"{task}_target has batch {Bn} for {B} predictions: the reference's repeat branch needs rank-3 targets "
                                     f"and B % Bn == 0 (base_ola_vlm.py:292-299)")
                rep = torch.empty(B, flat.shape[1], device=self.dev, dtype=BF16)
                key = ("tile_rows", Bn, B)
                if key not in self._static:
                    self._static[key] = torch.arange(B, device=self.dev, dtype=torch.int32) % Bn
                ops.gather_rows([flat], self._zeros_i32(B), self._static[key], flat.shape[1], rep)
                flat = rep
                mask = mask.repeat(B // Bn)
                Bn = B
            allt = all_gather_rows(flat, getattr(self, "comm", None),
                                   dry_world=self.world if getattr(self, "comm_dry", False) else 0) if self.world > 1 else flat
            if self.cfg.zero_masks:
                mask = torch.zeros_like(mask)
            out[task] = (allt, mask)
        return out

    # ------------------------------------------------------------------------------------------ one head
    def _head_fwd(self, task, i, idx, state, plan, batch, targets, compute_grads):
        """TaskToken{Gen,Seg,Depth}Head on one layer state: forward up to the loss input; returns the context _head_bwd continues from.
        resampler.py:202-224 / :46-75 / :9-16 ; gen_head.py:39-65 ; oneformer_head.py:224-258 ; da_v2_head.py:418-457."""
        cfg, ps, dev = self.cfg, self.ps, self.dev
        cname, _, wkey, sname, hname = TASK_SPEC[task]
        hc = getattr(cfg, cname)
        B, S, H = plan["B"], plan["S"], cfg.hidden_size
        tb = plan["heads"][task]
        n, nq, heads, dh = tb["n_x"], hc["num_tokens"], hc["num_heads"], hc["dim_head"]
        inner = heads * dh
        pf = f"{hname}.{i}.projector."
        T = n + nq
        # -- inputs: [x_b ; latents_b] per batch, contiguous (kv_input = cat(x, latents): resampler.py:59), built by ONE row gather from
        #    two sources: the tapped layer state and the latent source (the (576,H) task-token parameter for depth / seg; for gen the 8
        #    task-token rows of the state itself, or their per-sample mean when num_queries is not a multiple of 8: resampler.py:207-212)
        # kind / row tables of the [B*T] gather come with the splice plan (splice.head_tables: one pinned upload per batch, no sync)
        nl, mode, xt = tb["nl"], tb["mode"], tb["xin"]
        own = mode == "own"               # num_task_tokens == 0: plain Resampler with its own `latents` parameter (resampler.py:120-165)
        state2 = state.view(-1, H)
        Dm = ps.w(pf + "proj_in.weight").shape[0]
        if own:
            xin = torch.empty(B, n, H, device=dev, dtype=BF16)
            ops.gather_rows([state2], xt["kind"], xt["row"], H, xin)
            xin2 = xin.view(B * n, H)
            Px = ops.gemm(xin2, ps.w(pf + "proj_in.weight"), bias=ps.w(pf + "proj_in.bias"))            # proj_in on x only (:154)
            lat = torch.empty(B * nq, Dm, device=dev, dtype=BF16)                                        # latents.repeat(B, 1, 1) (:152)
            ops.gather_rows([ps.w(pf + "latents").view(nq, Dm)], self._zeros_i32(B * nq), xt["lat_rep"], Dm, lat)
        else:
            if task != "gen":
                lat_src = ps.w(f"model.special_{task}_tokens")
                if mode == "mean":                                                          # mean over the parameter rows (resampler.py:212)
                    lat_src = lat_src.float().mean(0, keepdim=True).to(BF16)
            elif mode == "mean":
                lat_src = torch.empty(B, H, device=dev, dtype=BF16)
                ops.gather_sum_rows(state2, xt["mean_idx"], nl, 1.0 / nl, lat_src)
            else:
                lat_src = state2
            xin = torch.empty(B, T, H, device=dev, dtype=BF16)
            ops.gather_rows([state2, lat_src], xt["kind"], xt["row"], H, xin)
            xin2 = xin.view(B * T, H)
            P = ops.gemm(xin2, ps.w(pf + "proj_in.weight"), bias=ps.w(pf + "proj_in.bias")).view(B, T, Dm)
            Px = torch.empty(B * n, Dm, device=dev, dtype=BF16)
            lat = torch.empty(B * nq, Dm, device=dev, dtype=BF16)
            ops.copy2d_(Px.view(B, n * Dm), P.view(B, T * Dm)[:, :n * Dm])
            ops.copy2d_(lat.view(B, nq * Dm), P.view(B, T * Dm)[:, n * Dm:])
        # -- `depth` Perceiver blocks (resampler.py:217-219): latents = attn(x, latents) + latents; latents = ff(latents) + latents.
        #    x (the projected token rows) is the same for every block; each block has its own norm1 / norm2 / projections.
        blocks = []
        for dpt in range(int(hc["depth"])):
            a, f = f"{pf}layers.{dpt}.0.", f"{pf}layers.{dpt}.1."
            Nn = torch.empty(B, T, Dm, device=dev, dtype=BF16)
            nx, mx_, rx = ops.layernorm_fwd(Px, ps.w(a + "norm1.weight"), ps.w(a + "norm1.bias"))
            nlat, ml_, rl = ops.layernorm_fwd(lat, ps.w(a + "norm2.weight"), ps.w(a + "norm2.bias"))
            ops.copy2d_(Nn.view(B, T * Dm)[:, :n * Dm], nx.view(B, n * Dm))
            ops.copy2d_(Nn.view(B, T * Dm)[:, n * Dm:], nlat.view(B, nq * Dm))
            wqkv = ps.fused([a + "to_q.weight", a + "to_kv.weight"])                               # [3*inner, Dm]: adjacent in the flat store
            QKV = ops.gemm(Nn.view(B * T, Dm), wqkv).view(B, T, 3 * inner)
            q4 = QKV[:, n:, :inner].unflatten(-1, (heads, dh))
            k4 = QKV[:, :, inner:2 * inner].unflatten(-1, (heads, dh))
            v4 = QKV[:, :, 2 * inner:].unflatten(-1, (heads, dh))
            att, lse = ops.attn_fwd(q4, k4, v4, causal=False, scale=1.0 / math.sqrt(dh))
            att2 = att.view(B * nq, inner)
            lat1 = ops.gemm(att2, ps.w(a + "to_out.weight"), residual=lat)
            y, mf, rf = ops.layernorm_fwd(lat1, ps.w(f + "0.weight"), ps.w(f + "0.bias"))
            zf = ops.gemm(y, ps.w(f + "1.weight"))
            af = ops.act_fwd(zf, ops.EPI_GELU)
            lat2 = ops.gemm(af, ps.w(f + "3.weight"), residual=lat1)
            if compute_grads:
                blocks.append((a, f, Nn, mx_, rx, lat, ml_, rl, wqkv, q4, k4, v4, att, lse, att2, lat1, y, mf, rf, zf, af))
            lat = lat2
        lat2 = lat
        po = ops.gemm(lat2, ps.w(pf + "proj_out.weight"), bias=ps.w(pf + "proj_out.bias"))
        vout, mo, ro = ops.layernorm_fwd(po, ps.w(pf + "norm_out.weight"), ps.w(pf + "norm_out.bias"))
        Do = vout.shape[-1]
        emb = vout.view(B, nq, Do)
        pred = vout
        uid = task == "depth" and bool(hc.get("use_intermediate_depth", True))      # base_ola_vlm.py:132; False: no linear_1..3, loss on visual_feats
        if uid:                                                    # loss on linear_1(visual_feats): base_ola_vlm.py:369
            l1 = f"{hname}.{i}.linear_1."
            zd = ops.gemm(vout, ps.w(l1 + "0.weight"), bias=ps.w(l1 + "0.bias"))
            ad = ops.act_fwd(zd, ops.EPI_RELU)
            pred = ops.gemm(ad, ps.w(l1 + "2.weight"), bias=ps.w(l1 + "2.bias"))
        res = dict(emb=emb if task != "depth" else pred.view(B, nq, -1), loss3=None, dx=None)
        if task == "depth" and getattr(cfg, "depth_decoder", False):
            # depth_embs entry = [lin1(v), lin2(v), lin3(v), v] (da_v2_head.py:444-457); depth_pred = DPT(feats) (base_ola_vlm.py:462-470);
            # without use_intermediate_depth: entry = [v], depth_pred = DPT([v] * 4) (da_v2_head.py:448-455, base_ola_vlm.py:464-465)
            if uid:
                fe = [pred]
                for j in (2, 3):
                    lj = f"{hname}.{i}.linear_{j}."
                    zz = ops.gemm(vout, ps.w(lj + "0.weight"), bias=ps.w(lj + "0.bias"), epi=ops.EPI_RELU)
                    fe.append(ops.gemm(zz, ps.w(lj + "2.weight"), bias=ps.w(lj + "2.bias")))
                fe.append(vout)
            else:
                fe = [vout]
            res["depth_feats"] = [t.view(B, nq, -1) for t in fe]
            res["depth_pred"] = self.dpt_forward(res["depth_feats"] if uid else res["depth_feats"] * 4)
        tg = targets.get(task)
        sname_ = sname
        scale = ps.p(sname_) if (cfg.use_contrastive and sname_ in ps) else None
        ctx = dict(task=task, i=i, idx=idx, res=res, tg=tg, pred2=pred.view(B, -1), scale=scale, w_t=float(hc[wkey]))
        if tg is not None and compute_grads:
            ctx["saved"] = dict(plan=plan, tb=tb, n=n, nq=nq, heads=heads, dh=dh, inner=inner, pf=pf, T=T, nl=nl, mode=mode, xt=xt, own=own,
                                Dm=Dm, xin2=xin2, Px=Px, blocks=blocks, lat2=lat2, po=po, mo=mo, ro=ro, vout=vout, hname=hname, sname=sname,
                                ad=ad if uid else None, zd=zd if uid else None, uid=uid)
        return ctx

    def _head_bwd(self, ctx):
        """Backward of one head from d(loss)/d(pred) (ctx["dpred"], from the batched loss backward) to the layer-state rows it read, its own
        parameters, the task-token / latent parameters and its logit scale."""
        cfg, ps, dev = self.cfg, self.ps, self.dev
        task, i, res = ctx["task"], ctx["i"], ctx["res"]
        sv = ctx["saved"]
        plan, tb, n, nq, heads, dh, inner, pf, T, nl, mode, xt, own = (sv[k] for k in ("plan", "tb", "n", "nq", "heads", "dh", "inner", "pf", "T", "nl", "mode", "xt", "own"))
        Dm, xin2, Px, blocks, lat2, po, mo, ro, vout, hname, sname, ad, zd = (sv[k] for k in ("Dm", "xin2", "Px", "blocks", "lat2", "po", "mo", "ro", "vout", "hname", "sname", "ad", "zd"))
        B, S, H = plan["B"], plan["S"], cfg.hidden_size
        l1 = f"{hname}.{i}.linear_1."
        w_t, coef, scale = ctx["w_t"], ctx["coef"], ctx["scale"]
        dpred = ctx["dpred"].view(B * nq, -1)
        if scale is not None:
            self._acc(ps.g(sname), coef[-1:], w_t)
        if sv["uid"]:
            d_ad = self._lin_bwd(ad, dpred, l1 + "2.weight", l1 + "2.bias")
            d_zd = ops.act_bwd(d_ad, zd, ops.EPI_RELU)
            dvout = self._lin_bwd(vout, d_zd, l1 + "0.weight", l1 + "0.bias")
        else:
            dvout = dpred
        d_po, _, _ = ops.layernorm_bwd(dvout, po, ps.w(pf + "norm_out.weight"), mo, ro, dw_out=ps.g(pf + "norm_out.weight"),
                                       db_out=ps.g(pf + "norm_out.bias"))
        d_lat = self._lin_bwd(lat2, d_po, pf + "proj_out.weight", pf + "proj_out.bias")       # gradient of the latents leaving the last block
        dPx = None
        for (a, f, Nn, mx_, rx, lat_in, ml_, rl, wqkv, q4, k4, v4, att, lse, att2, lat1, y, mf, rf, zf, af) in reversed(blocks):
            d_af = self._lin_bwd(af, d_lat, f + "3.weight")
            d_zf = ops.act_bwd(d_af, zf, ops.EPI_GELU)
            d_y = self._lin_bwd(y, d_zf, f + "1.weight")
            d_lat1, _, _ = ops.layernorm_bwd(d_y, lat1, ps.w(f + "0.weight"), mf, rf, dres=d_lat, dw_out=ps.g(f + "0.weight"),
                                             db_out=ps.g(f + "0.bias"))
            d_att = self._lin_bwd(att2, d_lat1, a + "to_out.weight")
            dQKV = ops.zero_(torch.empty(B, T, 3 * inner, device=dev, dtype=BF16))        # the x rows carry no queries: their dq stays 0
            ops.attn_bwd(q4, k4, v4, att, lse, d_att.view(B, nq, heads, dh), causal=False, scale=1.0 / math.sqrt(dh),
                         dq=dQKV[:, n:, :inner].unflatten(-1, (heads, dh)), dk=dQKV[:, :, inner:2 * inner].unflatten(-1, (heads, dh)),
                         dv=dQKV[:, :, 2 * inner:].unflatten(-1, (heads, dh)))
            dQKV2 = dQKV.view(B * T, 3 * inner)
            dNn = ops.gemm(dQKV2, self._wT(a + "to_q|to_kv", wqkv)).view(B, T, Dm)
            self._wgrad(Nn.view(B * T, Dm), dQKV2, ps.fused([a + "to_q.weight", a + "to_kv.weight"], ps.grad))
            dNx = torch.empty(B * n, Dm, device=dev, dtype=BF16)
            dNl = torch.empty(B * nq, Dm, device=dev, dtype=BF16)
            ops.copy2d_(dNx.view(B, n * Dm), dNn.view(B, T * Dm)[:, :n * Dm])
            ops.copy2d_(dNl.view(B, nq * Dm), dNn.view(B, T * Dm)[:, n * Dm:])
            dpx, _, _ = ops.layernorm_bwd(dNx, Px, ps.w(a + "norm1.weight"), mx_, rx, dres=dPx, dw_out=ps.g(a + "norm1.weight"),
                                          db_out=ps.g(a + "norm1.bias"))
            dPx = dpx                                             # every block reads the same projected tokens: their gradients add up
            d_lat, _, _ = ops.layernorm_bwd(dNl, lat_in, ps.w(a + "norm2.weight"), ml_, rl, dres=d_lat1,     # + residual path lat1 = to_out(.) + latents
                                            dw_out=ps.g(a + "norm2.weight"), db_out=ps.g(a + "norm2.bias"))
        dPl = d_lat
        if own:
            # the latents parameter was tiled over the batch: its gradient is the batch sum of the final latent gradients
            ops.gather_sum_rows(dPl, xt["lat_bwd"], xt["lat_cnt"], 1.0, ps.g(pf + "latents").view(nq, Dm), accumulate=True)
            dxg = self._lin_bwd(xin2, dPx, pf + "proj_in.weight", pf + "proj_in.bias").view(B, n, H)
            res["dx"] = dxg.view(B * n, H)
            return res
        dP = torch.empty(B, T, Dm, device=dev, dtype=BF16)
        ops.copy2d_(dP.view(B, T * Dm)[:, :n * Dm], dPx.view(B, n * Dm))
        ops.copy2d_(dP.view(B, T * Dm)[:, n * Dm:], dPl.view(B, nq * Dm))
        dxin = self._lin_bwd(xin2, dP.view(B * T, Dm), pf + "proj_in.weight", pf + "proj_in.bias").view(B, T, H)
        dxg = dxin[:, :n].contiguous()                                                  # grads of the gathered state rows
        # latents' gradient: the transpose of the forward row gather -> the (576,H) parameter (depth / seg) or the 8 hidden rows (gen)
        dxin2 = dxin.view(B * T, H)
        if task != "gen":
            gpar = ps.g(f"model.special_{task}_tokens")
            if mode == "mean":                                                          # every parameter row gets sum(dlat) / nl
                tot = dxin[:, n:].float().sum((0, 1)) / nl
                gpar.add_(tot[None, :].expand_as(gpar))
            else:
                ops.gather_sum_rows(dxin2, xt["lat_bwd"], xt["lat_cnt"], 1.0, gpar, accumulate=True)
        else:
            lo = int(tb["lat_x"][0])
            dlat_in = dxin[:, n:]
            if mode == "same":
                dlat0 = dlat_in.float()
            elif mode == "tile":
                dlat0 = dlat_in.float().reshape(B, nq // nl, nl, H).sum(1)
            else:
                dlat0 = (dlat_in.float().sum(1, keepdim=True) / nl).expand(B, nl, H)
            dxg[:, lo:lo + nl] = (dxg[:, lo:lo + nl].float() + dlat0).to(BF16)
        res["dx"] = dxg.view(B * n, H)
        return res
