"""Golden-vector generator: imports the REFERENCE ITSELF (read-only /root/reference) in the build
container and writes small input/output fixtures to tests/golden/.  TEST INFRASTRUCTURE; runs only
where /root/reference exists (never on the GPU box).  Nothing from the reference is copied: the
fixtures hold tensors (subsampled outputs, losses, gradients, a parameter-name manifest) only.

Recipe = SURVEY.md §8(c): third-party stubs, package shells that skip the reference __init__.py
files, a stand-in BaseCausalLM, world_size-1 gloo group, synthetic frozen-teacher targets.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types
import json

sys.dont_write_bytecode = True
import numpy as np
import torch
import transformers  # noqa: F401  (must precede the stubs)
from transformers import PreTrainedModel, CLIPVisionModel, CLIPVisionConfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import weights as WT  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return sys.modules.get(f"{self.__name__}.{n}") or type(n, (), {})


def _stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        k = ".".join(parts[:i])
        if k not in sys.modules:
            m = _Stub(k)
            m.__path__ = []
            sys.modules[k] = m


def setup_reference():
    sys.path.insert(0, REF)
    for n in ["open_clip.model", "open_clip.coca_model", "open_clip.openai", "open_clip.pretrained",
              "open_clip.transform", "open_clip.factory", "open_clip.transformer", "timm.models.convnext",
              "diffusers", "cv2", "diffdist.functional", "torchvision.ops", "torchvision.transforms", "wandb",
              "matplotlib", "matplotlib.pyplot", "matplotlib.cm", "icecream"]:
        try:
            __import__(n)
        except Exception:
            _stub(n)
    sys.modules["diffdist.functional"].all_gather = lambda out, x: (out.__setitem__(0, x) or out)

    def shell(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    shell("ola_vlm", f"{REF}/ola_vlm")
    shell("ola_vlm.model", f"{REF}/ola_vlm/model")
    shell("ola_vlm.model.language_model", f"{REF}/ola_vlm/model/language_model")
    bl = types.ModuleType("ola_vlm.model.language_model.base_lm")

    class BaseCausalLM(PreTrainedModel):
        def __init__(self, config):
            super().__init__(config)
    bl.BaseCausalLM = BaseCausalLM
    sys.modules[bl.__name__] = bl
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)


class KeepMask(torch.Tensor):
    """Mask whose in-place zero_() is a no-op -> the INTENDED (non-zeroed) loss."""
    def zero_(self):
        return self


def sub(t, n=4096):
    """Deterministic subsample of a tensor (flattened stride) to keep fixtures small."""
    f = t.detach().float().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def build_llama(tiny, arch="llama"):
    from ola_vlm.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from ola_vlm.model.multimodal_projector.builder import build_vision_projector
    if arch == "phi3":
        from ola_vlm.model.language_model.ola_phi3 import OlaLlavaPhi3ForCausalLM as OlaLlavaLlamaForCausalLM
        from ola_vlm.model.language_model.ola_phi3 import OlaLlavaPhi3Config as OlaLlavaLlamaConfig
        theta, extra = 10000.0, dict(sliding_window=tiny["sliding_window"], pad_token_id=0, resid_pdrop=0.0, embd_pdrop=0.0,
                                     attention_dropout=0.0)
    else:
        from ola_vlm.model.language_model.ola_llama import OlaLlavaLlamaForCausalLM, OlaLlavaLlamaConfig
        theta, extra = 500000.0, {}
    cfg = OlaLlavaLlamaConfig(vocab_size=tiny["vocab_size"], hidden_size=tiny["hidden_size"],
                              intermediate_size=tiny["intermediate_size"], num_hidden_layers=tiny["num_hidden_layers"],
                              num_attention_heads=tiny["num_attention_heads"],
                              num_key_value_heads=tiny["num_key_value_heads"], rms_norm_eps=1e-5,
                              max_position_embeddings=4096, **extra)
    try:
        cfg.rope_parameters = {"rope_type": "default", "rope_theta": theta}
    except Exception:
        pass
    cfg.rope_theta = theta
    cfg._attn_implementation = "eager"
    cfg.aux_mode = tiny["aux_mode"]
    cfg.num_task_tokens = tiny.get("num_task_tokens", 8)
    cfg.contrastive_loss_weight = 0.3
    cfg.image_gen = dict(tiny["image_gen"])
    cfg.image_seg = dict(tiny["image_seg"])
    cfg.image_depth = dict(tiny["image_depth"])
    cfg.image_generator = cfg.image_segmentor = "unused"
    p = "/tmp/_empty_sd.pth"
    torch.save({}, p)
    cfg.depth_estimator = p
    cfg.tokenizer_model_max_length = 4096
    cfg.tokenizer_padding_side = "right"
    model = OlaLlavaLlamaForCausalLM(cfg)
    tower = CLIPVisionTower.__new__(CLIPVisionTower)
    torch.nn.Module.__init__(tower)
    tower.is_loaded, tower.select_layer, tower.select_feature = True, -2, "patch"
    vcfg = CLIPVisionConfig(hidden_size=tiny["vit_hidden"], intermediate_size=tiny["vit_inter"],
                            num_hidden_layers=tiny["vit_layers"], num_attention_heads=tiny["vit_heads"],
                            image_size=336, patch_size=14)
    vcfg._attn_implementation = "eager"
    tower.vision_tower = CLIPVisionModel(vcfg).requires_grad_(False)
    model.model.vision_tower = tower
    cfg.mm_projector_type, cfg.mm_hidden_size = "mlp2x_gelu", tiny["vit_hidden"]
    model.model.mm_projector = build_vision_projector(cfg)
    if cfg.num_task_tokens > 0:                         # ola_vlm_train.py:1239-1240
        model.model.initialize_special_tokens(cfg)
    else:
        model.model.num_task_tokens, model.model.task_token_format = 0, "emb"          # what initialize_special_tokens would record
        model.model.aux_tokens, model.model.token_order = cfg.aux_mode, cfg.aux_mode.split("-")
    return model, cfg


TINY_LLAMA = dict(
    vocab_size=128256, hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4,
    num_key_value_heads=2, vit_hidden=128, vit_inter=256, vit_layers=3, vit_heads=4, aux_mode="gen-depth-seg",
    image_gen=dict(depth=1, dim_head=32, num_heads=4, num_tokens=1, output_dim=1024, ff_mult=1,
                   img_layer_indices="4", img_loss_weight=0.5),
    image_seg=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1536, ff_mult=1,
                   seg_layer_indices="2-3", seg_loss_weight=0.5),
    image_depth=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1024, ff_mult=1,
                     depth_layer_indices="3", depth_loss_weight=0.5),
)


def make_batch(B, T, img_col, vocab_hi=1000):
    ids = (WT.unit_uniform("input_ids", B * T).reshape(B, T) * 0.28867 + 0.5) * vocab_hi
    ids = torch.from_numpy(np.clip(ids, 0, vocab_hi - 1).astype(np.int64))
    ids[:, img_col] = -200
    labels = ids.clone()
    labels[:, :img_col + 7] = -100
    images = WT.tensor("images", (B, 3, 336, 336))
    tg = WT.tensor("gen_target", (B, 1, 1024))
    td = WT.tensor("depth_target", (B, 576, 1024))
    ts = WT.tensor("seg_target", (B, 1536, 24, 24))
    return ids, labels, images, tg, td, ts


TINY_PHI3 = dict(
    vocab_size=32064, hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4,
    num_key_value_heads=4, sliding_window=300, vit_hidden=128, vit_inter=256, vit_layers=3, vit_heads=4, aux_mode="gen-depth-seg",
    image_gen=dict(depth=1, dim_head=32, num_heads=4, num_tokens=1, output_dim=1024, ff_mult=1,
                   img_layer_indices="4", img_loss_weight=0.5),
    image_seg=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1536, ff_mult=1,
                   seg_layer_indices="2-3", seg_loss_weight=0.5),
    image_depth=dict(depth=1, dim_head=32, num_heads=4, num_tokens=576, output_dim=1024, ff_mult=1,
                     depth_layer_indices="3", depth_loss_weight=0.5),
)


def _fresh_tiny_llama(B, tiny=None, arch="llama"):
    """A NEW reference model per forward: under transformers 5.x (installed here; the reference pins
    4.41.1) the outer model's first `output_hidden_states=True` call leaves duplicate recorder hooks on
    the nested CLIP tower, so from the SECOND call on `hidden_states` has 2L+1 entries and
    `hidden_states[-2]` silently selects a different layer.  Only a first call has the pinned semantics."""
    model, cfg = build_llama(tiny or TINY_LLAMA, arch)
    # PT-stage trainability (ola_vlm_train.py:1127-1131,1147): LLM + tower frozen.
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: WT.param(k, s) for k, s in shapes.items()}
    model.load_state_dict(sd, strict=True)                 # including the frozen DPT decoder (da_v2_head.*)
    model.requires_grad_(False)
    for n, p in model.named_parameters():
        if ("mm_projector" in n or "_heads." in n or "special_" in n or "logit_scale" in n):
            p.requires_grad_(True)
    return model, shapes


def run_tiny_llama(arch="llama"):
    tiny = TINY_LLAMA if arch == "llama" else TINY_PHI3
    B, T, col = (2, 59, 38) if arch == "llama" else (2, 54, 13)
    ids, labels, images, tg, td, ts = make_batch(B, T, col)
    captured = []
    res = {}
    for mode in ("keep", "released"):
        model, shapes = _fresh_tiny_llama(B, tiny, arch)
        model._get_gen_feats = lambda pil, dev: tg
        model._get_seg_targets = lambda pil, h: ts
        model._get_dav2_feats = lambda pil, dev: ([(td, None)], torch.zeros(B, 336, 336))
        orig = model._emb_loss

        def spy(preds, mask, tgt, scale, orig=orig):
            r = orig(preds, mask, tgt, scale)
            captured.append((tuple(preds.shape), [float(x) for x in r], sub(preds, 512)))
            return r
        model._emb_loss = spy
        captured.clear()
        # released mode: a non-fp32 mask so `.float()` copies (an fp32 mask zeroed twice in place breaks the
        # reference's own backward when a task has two layers)
        mk = (lambda: torch.ones(B).as_subclass(KeepMask)) if mode == "keep" else (lambda: torch.ones(B, dtype=torch.float64))
        out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), labels=labels,
                    images=images, pil_images=[None] * B, gen_mask=mk(), seg_mask=mk(), depth_mask=mk())
        out.loss.backward()
        res[f"{mode}_loss"] = np.float64(out.loss.item())
        res[f"{mode}_layer_losses"] = np.array([c[1] for c in captured], dtype=np.float64)
        res[f"{mode}_layer_shapes"] = json.dumps([c[0] for c in captured])
        for j, c in enumerate(captured):
            res[f"{mode}_pred{j}"] = c[2]
        if mode == "keep":
            lg = out.logits
            res["logits_shape"] = np.array(lg.shape)
            res["logits_sub"] = lg[:, ::41, ::997].detach().numpy().copy()
            res["logits_lse_sub"] = torch.logsumexp(lg, -1)[:, ::7].detach().numpy().copy()
            hs = out.hidden_states
            res["n_hidden_states"] = np.array(len(hs))
            for li in (0, 2, 3, 4):
                res[f"hidden{li}_sub"] = hs[li][:, ::13, ::3].detach().numpy().copy()
            res["depth_embs_len"] = np.array(len(out.depth_embs[0]))
            res["depth_preds_shape"] = np.array(out.depth_preds[0].shape)
            res["depth_pred_sub"] = out.depth_preds[0][:, ::5, ::5].detach().float().numpy().copy()   # a11: DPT decoder output
            res["depth_pred_mean"] = np.float64(out.depth_preds[0].double().mean().item())
            res["seg_emb_sub"] = sub(out.seg_embs[0], 2048)
            res["gen_emb_sub"] = sub(out.image_embs[0], 1024)
        grads = {}
        for n, p in model.named_parameters():
            if p.requires_grad:
                grads[n] = None if p.grad is None else p.grad
        res[f"{mode}_grad_none"] = json.dumps(sorted(k for k, v in grads.items() if v is None))
        for n, g in grads.items():
            if g is not None:
                res[f"{mode}_gradnorm::{n}"] = np.float64(g.double().norm().item())
                if mode == "keep":
                    res[f"keep_gradsub::{n}"] = sub(g, 256)
    res["manifest"] = json.dumps({k: list(s) for k, s in shapes.items()})
    res["trainable"] = json.dumps(sorted(n for n, p in model.named_parameters() if p.requires_grad))
    res["input_ids"] = ids.numpy()
    res["labels"] = labels.numpy()
    res["cfg"] = json.dumps(tiny)
    res["batch"] = json.dumps([B, T, col])
    np.savez_compressed(os.path.join(OUT, f"tiny_{arch}_e2e.npz"), **res)
    print(f"tiny_{arch}: keep loss", res["keep_loss"], "released loss", res["released_loss"])
    print("layer losses (keep):\n", res["keep_layer_losses"], res["keep_layer_shapes"])
    return model


def run_tiny_ragged():
    """Padded rows (VERDICT r3 next-6 follow-up).  A right-padded batch: sample 1 is 17 tokens shorter (attention_mask False on its tail) in one
    run and carries NO <image> token in the other, so its tail of the spliced sequence is padding (ola_arch.py:337-338, 408-427).  The reference
    does not mask padded QUERY rows anywhere, their hidden states are real numbers (position ids: the caller passes none, so the reference hands
    `position_ids=None` on, ola_arch.py:439-440, and HF numbers every row of the padded tensor 0..S-1) and forward_emb_predictor feeds them to the
    heads of the shorter sample (base_ola_vlm.py:413-443).  The fixture pins those rows: final hidden states of EVERY row of sample 1, every
    layer's loss triple, and the gen / depth predictions per sample."""
    B, T, col = 2, 59, 38
    res = {}
    for tag in ("short", "noimg"):
        ids, labels, images, tg, td, ts = make_batch(B, T, col)
        am = torch.ones_like(ids, dtype=torch.bool)
        if tag == "short":
            am[1, 42:] = False
        else:
            ids[1, col] = 7
        model, shapes = _fresh_tiny_llama(B, TINY_LLAMA, "llama")
        model._get_gen_feats = lambda pil, dev: tg
        model._get_seg_targets = lambda pil, h: ts
        model._get_dav2_feats = lambda pil, dev: ([(td, None)], torch.zeros(B, 336, 336))
        captured = []
        orig = model._emb_loss

        def spy(preds, mask, tgt, scale, orig=orig):
            r = orig(preds, mask, tgt, scale)
            captured.append((tuple(preds.shape), [float(x) for x in r], preds.detach().float().reshape(preds.shape[0], -1)[:, ::max(1, preds[0].numel() // 509)].numpy().copy()))
            return r
        model._emb_loss = spy
        mk = lambda: torch.ones(B).as_subclass(KeepMask)
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=am, labels=labels, images=images, pil_images=[None] * B,
                        gen_mask=mk(), seg_mask=mk(), depth_mask=mk())
        hs = out.hidden_states
        res[f"{tag}_loss"] = np.float64(out.loss.item())
        res[f"{tag}_layer_losses"] = np.array([c[1] for c in captured], dtype=np.float64)
        res[f"{tag}_layer_shapes"] = json.dumps([c[0] for c in captured])
        for j, c in enumerate(captured):
            res[f"{tag}_pred{j}_sub"] = c[2]                                # [B, ~509 strided elements]: per sample
        res[f"{tag}_hidden_last_sample1"] = hs[-1][1, :, ::3].detach().numpy().copy()       # every row (real and padded), every 3rd feature
        res[f"{tag}_hidden2_sample1"] = hs[2][1, :, ::3].detach().numpy().copy()
        res[f"{tag}_input_ids"] = ids.numpy()
        res[f"{tag}_attention_mask"] = am.numpy()
        res[f"{tag}_labels"] = labels.numpy()
        print(f"tiny_ragged/{tag}: loss {res[f'{tag}_loss']:.6f}, S = {hs[-1].shape[1]}, layer losses\n", res[f"{tag}_layer_losses"])
    res["cfg"] = json.dumps(TINY_LLAMA)
    res["batch"] = json.dumps([B, T, col])
    np.savez_compressed(os.path.join(OUT, "tiny_llama_ragged.npz"), **res)


def run_tiny_ift(task_tokens=None):
    """IFT-stage golden: the reference's LlavaLlamaForCausalLM (llava_llama.py:50-119 + llava_arch.py:300-486, NTP loss
    only) with everything but the vision tower trainable (scripts/train/finetune.sh) -> loss + the norm and a subsample of EVERY
    parameter gradient.  task_tokens=None: no task tokens (a plain LLaVA checkpoint) -> tiny_llama_ift.npz.  task_tokens="emb": the
    reference's own PT -> IFT hand-off (a PT checkpoint's config: num_task_tokens=8, aux_mode gen-depth-seg, task_token_format "emb"):
    append_special_tokens splices the raw (576, H) depth / seg parameters + the 8 gen rows behind the image (llava_arch.py:250-293)
    -> tiny_llama_ift_tok.npz; the same call with task_token_format "text" is recorded as raising (embed_tokens on float parameters)."""
    from ola_vlm.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from ola_vlm.model.multimodal_projector.builder import build_vision_projector
    from ola_vlm.model.language_model.llava_llama import LlavaLlamaForCausalLM, LlavaConfig
    tiny = TINY_LLAMA
    cfg = LlavaConfig(vocab_size=tiny["vocab_size"], hidden_size=tiny["hidden_size"], intermediate_size=tiny["intermediate_size"],
                      num_hidden_layers=tiny["num_hidden_layers"], num_attention_heads=tiny["num_attention_heads"],
                      num_key_value_heads=tiny["num_key_value_heads"], rms_norm_eps=1e-5, max_position_embeddings=4096)
    try:
        cfg.rope_parameters = {"rope_type": "default", "rope_theta": 500000.0}
    except Exception:
        pass
    cfg.rope_theta = 500000.0
    cfg._attn_implementation = "eager"
    cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side = 4096, "right"
    if not hasattr(cfg, "pretraining_tp"):
        cfg.pretraining_tp = 1
    if task_tokens is not None:                         # what a PT-stage config.json carries (ola_vlm_train.py:1149-1157, 1237-1240)
        cfg.num_task_tokens, cfg.aux_mode, cfg.task_token_format = 8, tiny["aux_mode"], task_tokens
        cfg.image_depth, cfg.image_seg, cfg.image_gen = dict(tiny["image_depth"]), dict(tiny["image_seg"]), dict(tiny["image_gen"])

    def build(c):
        model = LlavaLlamaForCausalLM(c)
        tower = CLIPVisionTower.__new__(CLIPVisionTower)
        torch.nn.Module.__init__(tower)
        tower.is_loaded, tower.select_layer, tower.select_feature = True, -2, "patch"
        vcfg = CLIPVisionConfig(hidden_size=tiny["vit_hidden"], intermediate_size=tiny["vit_inter"], num_hidden_layers=tiny["vit_layers"],
                                num_attention_heads=tiny["vit_heads"], image_size=336, patch_size=14)
        vcfg._attn_implementation = "eager"
        tower.vision_tower = CLIPVisionModel(vcfg).requires_grad_(False)
        model.model.vision_tower = tower
        c.mm_projector_type, c.mm_hidden_size = "mlp2x_gelu", tiny["vit_hidden"]
        model.model.mm_projector = build_vision_projector(c)
        shapes_ = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: WT.param(k, s_) for k, s_ in shapes_.items()}, strict=True)
        model.requires_grad_(True)
        model.model.vision_tower.requires_grad_(False)
        return model, shapes_
    model, shapes = build(cfg)
    B, T, col = 2, 59, 38
    ids, labels, images, *_ = make_batch(B, T, col)
    out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), labels=labels, images=images)
    out.loss.backward()
    extra = {}
    if task_tokens is not None:
        import copy
        c2 = copy.deepcopy(cfg)
        c2.task_token_format = "text"
        m2, _ = build(c2)
        try:
            m2(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), labels=labels, images=images)
            extra["text_format_error"] = ""
        except Exception as e:                          # noqa: BLE001
            extra["text_format_error"] = f"{type(e).__name__}: {str(e)[:160]}"
        print("task_token_format='text' in the reference ->", extra["text_format_error"] or "ran")
        extra["task_token_format"] = task_tokens
        extra["cfg"] = json.dumps(dict(aux_mode=cfg.aux_mode, num_task_tokens=8, image_depth=cfg.image_depth, image_seg=cfg.image_seg,
                                       image_gen=cfg.image_gen))
    res = {**extra, "loss": np.float64(out.loss.item()), "logits_shape": np.array(out.logits.shape),
           "logits_sub": out.logits[:, ::41, ::997].detach().numpy().copy(),
           "manifest": json.dumps({k: list(s) for k, s in shapes.items()}),
           "trainable": json.dumps(sorted(n for n, p in model.named_parameters() if p.requires_grad)),
           "input_ids": ids.numpy(), "labels": labels.numpy(), "batch": json.dumps([B, T, col])}
    for n, p in model.named_parameters():
        if p.requires_grad:
            res[f"gradnorm::{n}"] = np.float64(p.grad.double().norm().item())
            res[f"gradsub::{n}"] = sub(p.grad, 128)
    name = "tiny_llama_ift.npz" if task_tokens is None else "tiny_llama_ift_tok.npz"
    np.savez_compressed(os.path.join(OUT, name), **res)
    print(name, ": loss", res["loss"], "logits", res["logits_shape"], "params with grad", len(json.loads(res["trainable"])))


def run_tiny_nt0():
    """num_task_tokens == 0 with distillation heads (VERDICT r2 missing-5): init_heads builds GenHead / DepthHead / OneFormerSegHead around
    the plain `Resampler` with its own `latents` parameter (base_ola_vlm.py:120-123,137-140,166-169; resampler.py:120-165), no
    special_*_tokens exist, and forward_emb_predictor hands the WHOLE layer state to the head (:420-422, :429-430).  The reference's own
    OlaLlavaLlamaForCausalLM -> tiny_llama_nt0.npz (loss, layer-loss triples, gradient norms of every trainable parameter)."""
    B, T, col = 2, 59, 38
    ids, labels, images, tg, td, ts = make_batch(B, T, col)
    tiny = dict(TINY_LLAMA, num_task_tokens=0)
    model, cfg = build_llama(tiny)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: WT.param(k, s) for k, s in shapes.items()}, strict=True)
    model.requires_grad_(False)
    for n, p in model.named_parameters():
        if ("mm_projector" in n or "_heads." in n or "special_" in n or "logit_scale" in n):
            p.requires_grad_(True)
    model._get_gen_feats = lambda pil, dev: tg
    model._get_seg_targets = lambda pil, h: ts
    model._get_dav2_feats = lambda pil, dev: ([(td, None)], torch.zeros(B, 336, 336))
    captured = []
    orig = model._emb_loss

    def spy(preds, mask, tgt, scale):
        r = orig(preds, mask, tgt, scale)
        captured.append((tuple(preds.shape), [float(x) for x in r]))
        return r
    model._emb_loss = spy
    mk = lambda: torch.ones(B).as_subclass(KeepMask)
    out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), labels=labels, images=images,
                pil_images=[None] * B, gen_mask=mk(), seg_mask=mk(), depth_mask=mk())
    out.loss.backward()
    res = {"loss": np.float64(out.loss.item()), "layer_losses": np.array([c[1] for c in captured], dtype=np.float64),
           "layer_shapes": json.dumps([c[0] for c in captured]), "logits_shape": np.array(out.logits.shape),
           "manifest": json.dumps({k: list(s) for k, s in shapes.items()}),
           "trainable": json.dumps(sorted(n for n, p in model.named_parameters() if p.requires_grad)),
           "input_ids": ids.numpy(), "batch": json.dumps([B, T, col]), "cfg": json.dumps(tiny),
           "seg_emb_sub": sub(out.seg_embs[0], 2048), "gen_emb_sub": sub(out.image_embs[0], 1024)}
    none = []
    for n, p in model.named_parameters():
        if p.requires_grad:
            if p.grad is None:
                none.append(n)
            else:
                res[f"gradnorm::{n}"] = np.float64(p.grad.double().norm().item())
                res[f"gradsub::{n}"] = sub(p.grad, 128)
    res["grad_none"] = json.dumps(sorted(none))
    np.savez_compressed(os.path.join(OUT, "tiny_llama_nt0.npz"), **res)
    print("tiny_llama_nt0: loss", res["loss"], "logits", res["logits_shape"], "layer losses\n", res["layer_losses"], res["layer_shapes"])


def run_tiny_noid():
    """image_depth["use_intermediate_depth"] = False (VERDICT r5 missing-2; base_ola_vlm.py:132,462-466, da_v2_head.py:437-455): the depth head
    has no linear_1..3, `features` = [(visual_feats, None)], the distillation loss compares visual_feats itself with the target and the frozen DPT
    decoder runs on [depth_feats[0]] * 4.  The reference's own OlaLlavaLlamaForCausalLM -> tiny_llama_noid.npz (loss, layer-loss triples,
    depth_embs length, depth_preds samples, gradient norms / samples of every trainable parameter)."""
    B, T, col = 2, 59, 38
    ids, labels, images, tg, td, ts = make_batch(B, T, col)
    tiny = dict(TINY_LLAMA, image_depth=dict(TINY_LLAMA["image_depth"], use_intermediate_depth=False))
    model, cfg = build_llama(tiny)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert not any(".linear_" in k for k in shapes), "the reference builds linear_1..3 only with use_intermediate_depth"
    model.load_state_dict({k: WT.param(k, s) for k, s in shapes.items()}, strict=True)
    model.requires_grad_(False)
    for n, p in model.named_parameters():
        if ("mm_projector" in n or "_heads." in n or "special_" in n or "logit_scale" in n):
            p.requires_grad_(True)
    model._get_gen_feats = lambda pil, dev: tg
    model._get_seg_targets = lambda pil, h: ts
    model._get_dav2_feats = lambda pil, dev: ([(td, None)], torch.zeros(B, 336, 336))
    captured = []
    orig = model._emb_loss

    def spy(preds, mask, tgt, scale):
        r = orig(preds, mask, tgt, scale)
        captured.append((tuple(preds.shape), [float(x) for x in r]))
        return r
    model._emb_loss = spy
    mk = lambda: torch.ones(B).as_subclass(KeepMask)
    out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), labels=labels, images=images,
                pil_images=[None] * B, gen_mask=mk(), seg_mask=mk(), depth_mask=mk())
    out.loss.backward()
    res = {"loss": np.float64(out.loss.item()), "layer_losses": np.array([c[1] for c in captured], dtype=np.float64),
           "layer_shapes": json.dumps([c[0] for c in captured]), "logits_shape": np.array(out.logits.shape),
           "manifest": json.dumps({k: list(s) for k, s in shapes.items()}),
           "trainable": json.dumps(sorted(n for n, p in model.named_parameters() if p.requires_grad)),
           "input_ids": ids.numpy(), "batch": json.dumps([B, T, col]), "cfg": json.dumps(tiny),
           "depth_embs_len": np.array(len(out.depth_embs[0])), "depth_preds_shape": np.array(out.depth_preds[0].shape),
           "depth_pred_sub": out.depth_preds[0][:, ::5, ::5].detach().float().numpy().copy(),
           "depth_emb_sub": sub(out.depth_embs[0][0][0], 2048)}
    none = []
    for n, p in model.named_parameters():
        if p.requires_grad:
            if p.grad is None:
                none.append(n)
            else:
                res[f"gradnorm::{n}"] = np.float64(p.grad.double().norm().item())
                res[f"gradsub::{n}"] = sub(p.grad, 128)
    res["grad_none"] = json.dumps(sorted(none))
    np.savez_compressed(os.path.join(OUT, "tiny_llama_noid.npz"), **res)
    print("tiny_llama_noid: loss", res["loss"], "depth_embs_len", res["depth_embs_len"], "layer losses\n", res["layer_losses"], res["layer_shapes"])


def run_data_path():
    """f-4: tokenizer_image_token / expand2square outputs of the reference's own functions (ola_vlm/mm_utils.py) on a toy
    whitespace tokenizer and synthetic PIL images."""
    from ola_vlm.mm_utils import tokenizer_image_token, expand2square
    from PIL import Image

    class Tok:
        bos_token_id = 1

        def __init__(self, bos):
            self.bos = bos

        def __call__(self, text):
            ids = [3 + (sum(map(ord, w)) % 997) for w in text.split()]
            return types.SimpleNamespace(input_ids=([1] if self.bos else []) + ids)
    prompts = ["<image>\na photo of a cat", "describe <image> and then <image> briefly", "no image here", "<image>", "", "tail image <image>"]
    res = {"prompts": prompts, "with_bos": [tokenizer_image_token(p, Tok(True)) for p in prompts],
           "no_bos": [tokenizer_image_token(p, Tok(False)) for p in prompts], "squares": []}
    for (w, h) in ((7, 3), (3, 8), (5, 5)):
        img = Image.fromarray((np.arange(w * h * 3).reshape(h, w, 3) % 251).astype(np.uint8), "RGB")
        sq = expand2square(img, (122, 116, 104))
        res["squares"].append({"size": [w, h], "out": np.asarray(sq).tolist()})
    # the supervised collator: ola_vlm/train/ola_vlm_train.py does not import under the installed transformers (Trainer-side private APIs), so
    # the class definition alone is compiled from the reference file (read in place, at generation time only) and run as is
    import ast
    import dataclasses
    import typing
    import transformers
    ref_file = os.path.join(REF, "ola_vlm", "train", "ola_vlm_train.py")
    tree = ast.parse(open(ref_file).read(), filename=ref_file)
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DataCollatorForSupervisedDataset"][0]
    ns = {"torch": torch, "transformers": transformers, "dataclass": dataclasses.dataclass, "Sequence": typing.Sequence, "Dict": typing.Dict,
          "IGNORE_INDEX": -100}
    exec(compile(ast.Module(body=[node], type_ignores=[]), ref_file, "exec"), ns)
    coll = ns["DataCollatorForSupervisedDataset"](tokenizer=types.SimpleNamespace(pad_token_id=0, model_max_length=9))
    cases = []
    for ragged in (False, True):
        inst = []
        for i, L in enumerate((5, 12, 9)):
            ids = torch.arange(1, L + 1) + 10 * i
            side = 5 if (ragged and i == 1) else 4
            inst.append(dict(input_ids=ids, labels=ids.clone().masked_fill(ids % 3 == 0, -100), image=torch.full((3, side, side), float(i)),
                             pil_image=None, seg_mask=int(i != 1), depth_mask=int(i == 1), gen_mask=1))
        b = coll(inst)
        cases.append({"ragged": ragged, "input_ids": b["input_ids"].tolist(), "labels": b["labels"].tolist(),
                      "attention_mask": b["attention_mask"].long().tolist(), "images_is_list": isinstance(b["images"], list),
                      "images_shape": None if isinstance(b["images"], list) else list(b["images"].shape),
                      "keys": sorted(b.keys()), "seg_mask": b["seg_mask"].tolist(), "depth_mask": b["depth_mask"].tolist(),
                      "gen_mask": b["gen_mask"].tolist(), "seg_mask_dtype": str(b["seg_mask"].dtype)})
    res["collator"] = cases
    # process_images (ola_vlm/mm_utils.py:309-333) through HF's own CLIPImageProcessor (the object clip_encoder.py:31 loads), built
    # offline with the openai/clip-vit-large-patch14-336 preprocessor settings; three synthetic PIL images (wide, tall, small square)
    from ola_vlm.mm_utils import process_images
    from transformers import CLIPImageProcessor
    ip = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, do_center_crop=True, do_normalize=True,
                            do_resize=True, do_rescale=True, do_convert_rgb=True, resample=3,
                            image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711])
    pis = {}
    for mode in ("pad", "square"):         # any other string takes the default branch (None would crash the reference: `"anyres_max" in None`)
        imgs = [Image.fromarray(((np.arange(w * h * 3).reshape(h, w, 3) * 7 + 13 * (np.arange(h)[:, None, None] % 5)) % 253).astype(np.uint8), "RGB")
                for (w, h) in ((90, 41), (37, 120), (24, 24))]
        px = process_images(imgs, ip, types.SimpleNamespace(image_aspect_ratio=mode))
        pis[str(mode)] = {"shape": list(px.shape), "sub": px[:, :, ::17, ::13].numpy().astype(np.float64).round(6).tolist(),
                          "mean": float(px.double().mean()), "absmax": float(px.abs().max())}
    res["process_images"] = pis
    with open(os.path.join(OUT, "data_path.json"), "w") as fh:
        json.dump(res, fh)
    print("data_path: ", res["with_bos"][1], "| collator keys", cases[0]["keys"])


def run_dinov2_teacher():
    """f-3: the reference's own DinoVisionTransformer (depth_anything_v2/dinov2.py) at reduced width/depth, pos-embed grid 37x37 (img_size
    518) evaluated at 336 px -> the depth teacher target of _get_dav2_feats: mean of 4 normed intermediate patch-token maps."""
    from functools import partial
    from ola_vlm.model.aux_heads.depth_anything_v2.dinov2 import DinoVisionTransformer
    from ola_vlm.model.aux_heads.depth_anything_v2.dinov2_layers.block import Block
    from ola_vlm.model.aux_heads.depth_anything_v2.dinov2_layers.attention import MemEffAttention
    dims = dict(embed_dim=128, depth=6, num_heads=4)
    taps = [1, 2, 4, 5]
    m = DinoVisionTransformer(img_size=518, patch_size=14, init_values=1.0, ffn_layer="mlp", block_chunks=0, num_register_tokens=0,
                              interpolate_antialias=False, interpolate_offset=0.1, mlp_ratio=4,
                              block_fn=partial(Block, attn_class=MemEffAttention), **dims).eval()
    shapes = {"dav2_backbone.pretrained." + k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: WT.param(k, s) for k, s in shapes.items()}
    for k in sd:                                               # layer scale / tokens large enough to matter
        if k.endswith(".gamma"):
            sd[k] = WT.tensor(k, shapes[k], 0.3, 1.0)
        if k.endswith("cls_token") or k.endswith("pos_embed") or k.endswith("mask_token"):
            sd[k] = WT.tensor(k, shapes[k], 0.2)
    m.load_state_dict({k[len("dav2_backbone.pretrained."):]: v for k, v in sd.items()}, strict=True)
    images = WT.tensor("dino_images", (2, 3, 336, 336))
    with torch.no_grad():
        feats = m.get_intermediate_layers(images, taps, return_class_token=True)
        tgt = (feats[0][0] + feats[1][0] + feats[2][0] + feats[3][0]) / 4
    np.savez_compressed(os.path.join(OUT, "dinov2_teacher.npz"), manifest=json.dumps({k: list(s) for k, s in shapes.items()}),
                        dims=json.dumps(dims), taps=np.array(taps), target_shape=np.array(tgt.shape), target_sub=tgt[:, ::7, ::3].numpy().copy(),
                        target_mean=np.float64(tgt.double().mean().item()), target_std=np.float64(tgt.double().std().item()))
    print("dinov2_teacher:", tuple(tgt.shape), float(tgt.std()))


def run_clip_embed_teacher():
    """f-3: the generation teacher = `pipe.image_encoder(x).image_embeds` (base_ola_vlm.py:323-332) with HF's own
    CLIPVisionModelWithProjection (the class diffusers' unCLIP pipeline holds) at reduced width / depth, gelu activation as in ViT-H."""
    from transformers import CLIPVisionModelWithProjection
    dims = dict(hidden_size=128, intermediate_size=320, num_hidden_layers=3, num_attention_heads=4, image_size=224, patch_size=14,
                projection_dim=96, hidden_act="gelu")
    vc = CLIPVisionConfig(**dims)
    vc._attn_implementation = "eager"
    m = CLIPVisionModelWithProjection(vc).eval()
    shapes = {"pipe.image_encoder." + k: tuple(v.shape) for k, v in m.state_dict().items() if "position_ids" not in k}
    m.load_state_dict({k[len("pipe.image_encoder."):]: WT.param(k, s) for k, s in shapes.items()}, strict=False)
    images = WT.tensor("clip_embed_images", (2, 3, 224, 224))
    with torch.no_grad():
        emb = m(images).image_embeds.unsqueeze(1)
    np.savez_compressed(os.path.join(OUT, "clip_embed_teacher.npz"), manifest=json.dumps({k: list(s) for k, s in shapes.items()}),
                        dims=json.dumps(dims), embeds=emb.numpy().copy())
    print("clip_embed_teacher:", tuple(emb.shape), float(emb.std()))


def run_swin_teacher():
    """f-3: the segmentation teacher = Swin backbone feature_maps[-1] resized to 24 x 24 (oneformer_head.py:11-69) with HF's own SwinBackbone
    (what OneFormer's pixel-level module holds) at reduced width / depth: window 12, head_dim 32, 384-px input (grids 96/48/24/12)."""
    from transformers import SwinConfig, SwinBackbone
    dims = dict(image_size=384, patch_size=4, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=12)
    cfg = SwinConfig(**dims, out_features=["stage1", "stage2", "stage3", "stage4"])
    cfg._attn_implementation = "eager"
    m = SwinBackbone(cfg).eval()
    pre = "oneformer.model.pixel_level_module.encoder."
    shapes = {pre + k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: WT.param(k, s) for k, s in shapes.items()}
    for k in sd:
        if k.endswith("relative_position_bias_table"):
            sd[k] = WT.tensor(k, shapes[k], 0.5)                 # biases large enough to matter
    m.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=True)
    images = WT.tensor("swin_images", (2, 3, 384, 384))
    with torch.no_grad():
        fm = m(images).feature_maps[-1]
        tgt = torch.nn.functional.interpolate(fm, size=(24, 24), mode="bilinear", align_corners=False)
    np.savez_compressed(os.path.join(OUT, "swin_teacher.npz"), manifest=json.dumps({k: list(s) for k, s in shapes.items()}),
                        dims=json.dumps(dims), target_shape=np.array(tgt.shape), target_sub=tgt[:, ::5, ::3, ::3].numpy().copy(),
                        target_mean=np.float64(tgt.double().mean().item()), target_std=np.float64(tgt.double().std().item()))
    print("swin_teacher:", tuple(fm.shape), tuple(tgt.shape), float(tgt.std()))


def run_units():
    """Unit fixtures straight from the reference functions."""
    from ola_vlm.ola_utils import calculate_contrastive_loss
    from ola_vlm.model.language_model.base_ola_vlm import BaseOLA_VLM
    from ola_vlm.model.multimodal_projector.resampler import TaskTokenResampler
    res = {}
    fake = types.SimpleNamespace(contrastive_loss_weight=0.3)
    cases = {"gen": (3, 1, 1024), "depth": (3, 40, 1024), "seg": (3, 96, 6, 6)}
    for name, shp in cases.items():
        p = WT.tensor(f"unit_pred_{name}", shp, 1.3).requires_grad_(True)
        t = WT.tensor(f"unit_tgt_{name}", shp, 1.0)
        mask = torch.tensor([1.0, 0.0, 1.0])
        s = torch.tensor(2.0, requires_grad=True)
        e, l1, c = BaseOLA_VLM._emb_loss(fake, p, mask, t, s)
        e.backward()
        res[f"{name}_out"] = np.array([e.item(), l1.item(), c.item()], dtype=np.float64)
        res[f"{name}_dpred"] = p.grad.numpy().copy()
        res[f"{name}_dscale"] = np.float64(s.grad.item())
        res[f"{name}_con"] = calculate_contrastive_loss(p.detach(), t, s.detach()).numpy().copy()
    # saturating logit scale (clamp at 100) and no-contrastive variant
    p = WT.tensor("unit_pred_sat", (4, 8, 16), 1.0)
    t = WT.tensor("unit_tgt_sat", (4, 8, 16), 1.0)
    res["sat_con"] = calculate_contrastive_loss(p, t, torch.tensor(5.0)).numpy().copy()
    e, l1, c = BaseOLA_VLM._emb_loss(fake, p, torch.ones(4), t, None)
    res["nocon_out"] = np.array([float(e), float(l1), float(c)], dtype=np.float64)
    # TaskTokenResampler: gen-like (1 query from 8 latents) and seg-like (16 queries tiled from 8... uses mean path)
    for name, (dim, nq, emb, out_dim, nlat) in {"rs_gen": (64, 1, 48, 64, 8), "rs_tile": (32, 16, 48, 40, 8),
                                               "rs_same": (32, 12, 48, 40, 12), "rs_mean": (32, 6, 48, 40, 4),
                                               "rs_deep": (32, 12, 48, 40, 12)}.items():        # rs_deep: depth = 2 (resampler.py:217)
        m = TaskTokenResampler(dim=dim, depth=2 if name == "rs_deep" else 1, dim_head=32, num_queries=nq, heads=4, embedding_dim=emb,
                               output_dim=out_dim, ff_mult=1)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: WT.param(f"{name}.{k}", s) for k, s in shapes.items()})
        x = WT.tensor(f"{name}.x", (2, 50, emb))
        lat = WT.tensor(f"{name}.lat", (2, nlat, emb))
        res[f"{name}_out"] = m(x, lat).detach().numpy().copy()
        res[f"{name}_manifest"] = json.dumps({k: list(s) for k, s in shapes.items()})
    # plain Resampler (num_task_tokens == 0 heads; resampler.py:120-165): own latents, proj_in on x only; depth 1 and 2
    from ola_vlm.model.multimodal_projector.resampler import Resampler
    for name, (dim, nq, emb, out_dim, depth) in {"rs_plain": (32, 12, 48, 40, 1), "rs_plain_deep": (64, 5, 48, 24, 2)}.items():
        m = Resampler(dim=dim, depth=depth, dim_head=32, heads=4, num_queries=nq, embedding_dim=emb, output_dim=out_dim, ff_mult=1)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: WT.param(f"{name}.{k}", s) for k, s in shapes.items()})
        x = WT.tensor(f"{name}.x", (2, 50, emb))
        res[f"{name}_out"] = m(x).detach().numpy().copy()
        res[f"{name}_manifest"] = json.dumps({k: list(s) for k, s in shapes.items()})
    # _emb_loss's batch-repeat branch (base_ola_vlm.py:292-299): 4 predictions against 2 targets / 2 mask entries (rank-3 targets)
    for name, shp in {"rep_gen": (1, 1024), "rep_depth": (40, 256)}.items():
        p = WT.tensor(f"unit_pred_{name}", (4, *shp), 1.3).requires_grad_(True)
        t = WT.tensor(f"unit_tgt_{name}", (2, *shp), 1.0)
        s = torch.tensor(2.0, requires_grad=True)
        e, l1, c = BaseOLA_VLM._emb_loss(fake, p, torch.tensor([1.0, 0.5]), t, s)
        e.backward()
        res[f"{name}_out"] = np.array([e.item(), l1.item(), c.item()], dtype=np.float64)
        res[f"{name}_dpred"] = p.grad.numpy().copy()
        res[f"{name}_dscale"] = np.float64(s.grad.item())
    try:                                                # the same branch on a rank-4 (seg) target: the reference's 3-argument repeat raises
        BaseOLA_VLM._emb_loss(fake, WT.tensor("unit_pred_rep4", (4, 8, 3, 3)), torch.ones(2), WT.tensor("unit_tgt_rep4", (2, 8, 3, 3)), None)
        res["rep_rank4_error"] = ""
    except Exception as e:                              # noqa: BLE001
        res["rep_rank4_error"] = f"{type(e).__name__}: {str(e)[:120]}"
    np.savez_compressed(os.path.join(OUT, "units.npz"), **res)
    print("units written:", sorted(res)[:8], "...")


CNX_PREFIX = "model.vision_tower.vision_tower."


def cnx_timm_to_hf(name):
    """timm ConvNeXt state-dict name (what open_clip's `visual.trunk` holds, and what this repo's checkpoints / oracle use) -> the name of the
    same tensor in transformers.ConvNextModel, an independent third-party implementation of the published ConvNeXt block."""
    parts = name.split(".")
    if parts[0] == "stem":
        return ("embeddings.patch_embeddings." if parts[1] == "0" else "embeddings.layernorm.") + parts[2]
    assert parts[0] == "stages", name
    st = f"encoder.stages.{parts[1]}."
    if parts[2] == "downsample":
        return st + f"downsampling_layer.{parts[3]}.{parts[4]}"
    blk = st + f"layers.{parts[3]}."                       # stages.i.blocks.j.<rest>
    rest = ".".join(parts[4:])
    table = {"gamma": "layer_scale_parameter", "conv_dw.weight": "dwconv.weight", "conv_dw.bias": "dwconv.bias",
             "norm.weight": "layernorm.weight", "norm.bias": "layernorm.bias", "mlp.fc1.weight": "pwconv1.weight",
             "mlp.fc1.bias": "pwconv1.bias", "mlp.fc2.weight": "pwconv2.weight", "mlp.fc2.bias": "pwconv2.bias"}
    return blk + table[rest]


def run_convnext():
    """Row a2 pin.  timm / open_clip are absent, so the reference's `CLIPConvNextVisionTower._forward`
    (clip_convnext_encoder.py:150-174 — it only touches `vision_tower.stem`, `.stages`, `.norm_pre`) is run HERE on a stand-in trunk assembled from
    the sub-modules of transformers.ConvNextModel: an implementation of the published ConvNeXt block written by other people than timm's and
    than this repo's.  Weights: the closed-form recipe by timm name, loaded into the HF modules through the name map above.  The fixture holds
    the reference call's output, the name map and the dims — tensors and strings only."""
    from transformers import ConvNextConfig, ConvNextModel
    from ola_vlm.model.multimodal_encoder.clip_convnext_encoder import CLIPConvNextVisionTower
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.params import param_shapes
    dims, depths, px = (32, 64, 96, 128), (1, 2, 3, 1), 160                       # 160 / 32 = 5 -> 25 tokens
    hf = ConvNextModel(ConvNextConfig(num_channels=3, patch_size=4, num_stages=4, hidden_sizes=list(dims), depths=list(depths),
                                      hidden_act="gelu", layer_scale_init_value=1.0)).eval()
    vc = VisperConfig(mm_vision_tower="CLIP-convnext-pin", cnx_dims=dims, cnx_depths=depths)
    shapes = {k[len(CNX_PREFIX):]: tuple(v) for k, v in param_shapes(vc).items() if k.startswith(CNX_PREFIX)}
    sd = hf.state_dict()
    name_map, new_sd = {}, {}
    for k, shp in shapes.items():
        h = cnx_timm_to_hf(k)
        assert h in sd and tuple(sd[h].shape) == shp, (k, h, shp)
        name_map[k] = h
        new_sd[h] = WT.tensor(CNX_PREFIX + k, shp, 0.5) if k.endswith("gamma") else WT.param(CNX_PREFIX + k, shp)
    left = sorted(set(sd) - set(new_sd))
    assert left == ["layernorm.bias", "layernorm.weight"], left                   # HF's pooled-output norm: not on the reference's path
    hf.load_state_dict(new_sd, strict=False)

    class Trunk(torch.nn.Module):                                                  # the three attributes the reference touches, nothing else
        def __init__(self):
            super().__init__()
            self.stem, self.stages, self.norm_pre = hf.embeddings, hf.encoder.stages, torch.nn.Identity()
    tower = CLIPConvNextVisionTower.__new__(CLIPConvNextVisionTower)
    torch.nn.Module.__init__(tower)
    tower.vision_tower, tower.is_loaded, tower.is_multi_stage = Trunk(), True, False
    tower._image_size, tower._interp_size, tower._reduction = px, None, 32
    images = WT.tensor("cnx_pin_images", (2, 3, px, px))
    with torch.no_grad():
        feats = tower._forward(images)                                            # the reference's own code path
        stage_outs, x = [], hf.embeddings(images)
        for st in hf.encoder.stages:
            x = st(x)
            stage_outs.append(x)
    assert feats.shape == (2, (px // 32) ** 2, dims[-1]), feats.shape
    assert tower.num_patches == (px // 32) ** 2
    res = {"features": feats.numpy(), "dims": np.array(dims), "depths": np.array(depths), "px": np.array(px),
           "eps": np.array(1e-6), "name_map": np.array(json.dumps(name_map))}
    for i, so in enumerate(stage_outs):
        res[f"stage{i}_sub"] = sub(so)
        res[f"stage{i}_norm"] = np.array(float(so.norm()))
    np.savez_compressed(os.path.join(OUT, "convnext.npz"), **res)
    print("convnext written:", feats.shape, float(feats.abs().mean()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    setup_reference()
    which = sys.argv[1:] or ["units", "llama", "phi3", "ift", "ift_tok", "nt0", "noid", "data", "dino", "clipemb", "swin", "convnext", "ragged"]
    if "units" in which:
        run_units()
    if "llama" in which:
        run_tiny_llama("llama")
    if "phi3" in which:
        run_tiny_llama("phi3")
    if "ift" in which:
        run_tiny_ift()
    if "ift_tok" in which:
        run_tiny_ift("emb")
    if "nt0" in which:
        run_tiny_nt0()
    if "noid" in which:
        run_tiny_noid()
    if "data" in which:
        run_data_path()
    if "dino" in which:
        run_dinov2_teacher()
    if "clipemb" in which:
        run_clip_embed_teacher()
    if "swin" in which:
        run_swin_teacher()
    if "convnext" in which:
        run_convnext()
    if "ragged" in which:
        run_tiny_ragged()
