"""Parity at REAL WIDTH (CLIP-ViT-L 1024-d, Llama-3-8B H=4096 / 32q-8kv x128 / FF 14336 / V 128256; Phi-3-mini H=3072 / 32 x 96 / FF 8192,
S=4096, sliding window) of the HIP step against the CPU oracle on identical random-init weights and synthetic batches:

  * configs[0]-shaped (B=2 images, text 128 -> S=727 with three task-token groups) step with 2 decoder layers and ALL three
    full-width heads (seg D=1536, depth D=4096, gen D=1024): oracle in fp32 math on the same bf16-rounded weights/inputs -> isolates
    kernel error; losses, per-layer embedding losses, hidden states, logits rows and every trainable gradient.
  * BASELINE configs[0] itself (32 layers, seg@18, B=2, T=128) against the oracle run the way the reference runs on CPU (bf16
    weights, bf16 PyTorch ops) -> the north-star's "matches the reference CPU PyTorch path within 1e-3 bf16 relative".
  * configs[4]-shaped Phi-3 step (B=1, S=4096 > sliding window 2047, D=96, fused qkv/gate_up weights), 2 layers, fp32-math oracle.
Bounds are measured error x ~3 (tests/parity.py logs the measured values on every run)."""
import json
import os
import time

import pytest
import torch

from parity import check, rel, max_rel, grad_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _weights(cfg, seed=0):
    """Random-init state dict generated on the GPU (params.init_value: the recipe bench.py uses), bf16."""
    from visper_lm_amd.params import param_shapes, init_value
    gen = torch.Generator(device="cuda").manual_seed(seed)
    return {k: init_value(k, s, gen, torch.device("cuda"), BF) for k, s in param_shapes(cfg, vit_nested=True).items()}


def _batch(cfg, B, T, seed=7):
    g = torch.Generator().manual_seed(seed)
    ns = cfg.num_sys_tokens
    ids = torch.randint(0, 1000, (B, T), generator=g)
    ids[:, ns] = -200
    labels = ids.clone()
    labels[:, :ns + 7] = -100
    rn = lambda *s: torch.randn(*s, generator=g).to(BF)
    b = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids, dtype=torch.bool), images=rn(B, 3, 336, 336))
    if "gen" in cfg.token_order:
        b["gen_target"], b["gen_mask"] = rn(B, 1, cfg.image_gen["output_dim"]), torch.ones(B)
    if "depth" in cfg.token_order:
        b["depth_target"], b["depth_mask"] = rn(B, 576, cfg.image_depth["output_dim"]), torch.ones(B)
    if "seg" in cfg.token_order:
        b["seg_target"], b["seg_mask"] = rn(B, cfg.image_seg["output_dim"], 24, 24), torch.ones(B)
    return b


def _gpu(batch):
    return {k: (v.cuda() if (k == "images" or k.endswith("_target") or k.endswith("_mask")) else v) for k, v in batch.items()}


def _frob(got, want):
    """relative Frobenius error ||got - want|| / ||want||."""
    g, w = got.double(), want.double()
    return float((g - w).norm() / w.norm().clamp_min(1e-300))


def _hip_step(cfg, B, T):
    from visper_lm_amd.engine import Engine, is_trainable
    W = _weights(cfg)
    batch = _batch(cfg, B, T)
    eng = Engine(cfg)
    eng.load_weights(W)
    eng.keep_logits = True
    out = eng.train_step(_gpu(batch))
    torch.cuda.synchronize()
    S = out["plan"]["S"]
    rows = torch.linspace(0, S - 1, 64).long()
    got = dict(S=S, rows=rows, text_loss=float(out["text_loss"]), loss=float(out["loss"]),
               layer_losses={k: v.float().cpu() for k, v in out["layer_losses"].items()},
               inputs_embeds=out["inputs_embeds"].float().cpu(), hidden=out["hidden"].float().cpu(),
               logits=out["logits"][:, rows].float().cpu(), grads={k: eng.ps.g(k).detach().float().cpu().clone() for k in eng.ps.index})
    Wc = {k: v.detach().cpu() for k, v in W.items() if not k.startswith("da_v2_head.")}
    tr = [k for k in Wc if is_trainable(k)]
    del eng, out, W
    torch.cuda.empty_cache()
    return got, Wc, batch, tr


def _oracle_step(cfg, Wc, batch, tr, dtype, rows):
    from oracle import visper_oracle as O
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ocfg = O.make_config(**{k: v for k, v in cfg.to_dict().items() if k in vars(O.make_config())})
    Wo = {}
    for k, v in Wc.items():
        t = v.float() if (dtype == torch.float32 or v.dim() == 0) else v
        Wo[k] = t.clone().requires_grad_(True) if k in tr else t
    bo = {k: (v.to(dtype) if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    t0 = time.time()
    ref = O.forward(Wo, bo, ocfg)
    ref["loss"].backward()
    res = dict(text_loss=float(ref["text_loss"]), loss=float(ref["loss"]), seconds=time.time() - t0,
               layer_losses={k: torch.tensor([float(x) for x in v]) for k, v in ref["layer_losses"].items()},
               inputs_embeds=ref["inputs_embeds"].detach().float(), hidden=ref["hidden"].detach().float(),
               logits=ref["logits"].detach()[:, rows].float(),
               grads={k: (None if Wo[k].grad is None else Wo[k].grad.detach().float()) for k in tr})
    return res


def _compare(tag, got, ref, bounds, baseline=None):
    """Every metric is logged (parity.check); `bounds` may omit a key to log without asserting (bound = inf).  `baseline` = the
    gradient errors of the reference-style bf16 CPU path against the same fp32 truth (returned by an earlier _compare): a gradient
    may then deviate by bounds["grad_vs_baseline"] x the bf16 CPU path's own deviation (never less than the fixed bound).  Why:
    through ReLU gates (depth head linear_1) and LayerNorm backward, bf16 rounding of the activations moves some weight gradients
    by tens of percent relative to fp32 arithmetic in ANY bf16 implementation, the reference's included."""
    bd = lambda k: bounds.get(k, float("inf"))
    errs = {}
    check(f"{tag}/text_loss_rel", rel(got["text_loss"], ref["text_loss"]), bd("loss"))
    check(f"{tag}/loss_rel", rel(got["loss"], ref["loss"]), bd("loss"))
    for key, trip in ref["layer_losses"].items():
        mine = got["layer_losses"][key]
        for j, nm in enumerate(("emb", "sl1", "con")):
            if abs(float(trip[j])) > 1e-9:
                check(f"{tag}/layer_loss/{key[0]}@{key[1]}/{nm}_rel", rel(mine[j], trip[j]), bd("layer_loss"))
    state = {}
    for nm in ("inputs_embeds", "hidden", "logits"):
        state[nm + "_frob"], state[nm + "_max"] = _frob(got[nm], ref[nm]), max_rel(got[nm], ref[nm])
        # element-wise state bounds may be ADAPTIVE (round 6): bounds["state_vs_baseline"] = (factor, cap) -> min(cap, factor x the deviation of
        # the reference-style bf16 CPU path from the same truth), never looser than the cap
        fb, mb = bd(nm + "_frob"), bd(nm + "_max")
        if baseline is not None and "state_vs_baseline" in bounds and "__state__" in baseline:
            fac, cap = bounds["state_vs_baseline"]
            fb = min(fb, cap, fac * baseline["__state__"][nm + "_frob"])
            mb = min(mb, cap, fac * baseline["__state__"][nm + "_max"])
        check(f"{tag}/{nm}_frob", state[nm + "_frob"], fb)
        check(f"{tag}/{nm}_maxrel", state[nm + "_max"], mb)
    errs["__state__"] = state
    worst_c = worst_n = 0.0
    for k, want in ref["grads"].items():
        mine = got["grads"][k]
        if want is None:                                        # depth linear_2 / linear_3 only feed the no-grad DPT decoder
            if mine is not None:
                assert float(mine.abs().max()) == 0.0, k
            continue
        if mine is None:
            continue
        if want.numel() == 1:                                   # logit scales: a sum of signed terms -> absolute slack
            sb = (bd("grad_scalar_rel") * abs(float(want)) + bd("grad_scalar_abs")) if "grad_scalar_rel" in bounds else float("inf")
            sb = float("inf") if sb != sb else sb
            check(f"{tag}/grad/{k}_abs(ref {float(want):+.3e})", abs(float(mine) - float(want)), sb)
            continue
        c, n = grad_err(mine, want)
        errs[k] = (c, n)
        worst_c, worst_n = max(worst_c, c), max(worst_n, n)
        bc, bn = bd("grad_cos"), bd("grad_norm")
        if baseline is not None and k in baseline and k != "__state__" and "grad_vs_baseline" in bounds:
            bc = max(bc, bounds["grad_vs_baseline"] * baseline[k][0])
            bn = max(bn, bounds["grad_vs_baseline"] * baseline[k][1])
        check(f"{tag}/grad/{k}/one_minus_cos", c, bc)
        check(f"{tag}/grad/{k}/norm_dev", n, bn)
    print(f"[parity] {tag}: worst gradient 1-cos {worst_c:.2e}, worst norm deviation {worst_n:.2e}")
    return errs


def _run(cfg, B, T, tag, bounds):
    """HIP step vs the fp32-arithmetic oracle (the truth for these bf16 weights), with the reference-style bf16 CPU path's own
    deviation from that truth logged beside it and used as the yardstick for the gradients."""
    got, Wc, batch, tr = _hip_step(cfg, B, T)
    ref32 = _oracle_step(cfg, Wc, batch, tr, torch.float32, got["rows"])
    refb = _oracle_step(cfg, Wc, batch, tr, BF, got["rows"])
    print(f"[parity] {tag}: oracle fwd+bwd fp32 {ref32['seconds']:.1f} s, bf16 {refb['seconds']:.1f} s on {torch.get_num_threads()} threads, S={got['S']}")
    base = _compare(tag + "_INFO_bf16_cpu_path_vs_fp32_truth", refb, ref32, {})
    _compare(tag, got, ref32, bounds, baseline=base)
    return got, ref32


# measured (r02, parity_measured.jsonl): loss 1.7e-4, layer losses 1.0e-3, inputs_embeds 6.6e-3, hidden / logits 1.35e-2 (Frobenius) and
# 1.7e-2 (max); gradients: 1 - cos <= 3e-4 except the depth head (ReLU gates + LayerNorm backward amplify bf16 activation rounding: up to
# 4.5e-2 — and the bf16 CPU path shows 4.6e-2 on the same parameters, ratio HIP / CPU 0.97-1.01)
TIGHT = dict(loss=1e-3, layer_loss=5e-3, inputs_embeds_frob=2e-2, hidden_frob=4e-2, logits_frob=4e-2, inputs_embeds_max=2e-2,
             hidden_max=5e-2, logits_max=5e-2, grad_scalar_rel=0.05, grad_scalar_abs=3e-4, grad_cos=5e-3, grad_norm=2e-2,
             grad_vs_baseline=1.5)


def test_fullwidth_llama_step_all_heads_vs_fp32_oracle():
    """ola_llama.py:105-136 + base_ola_vlm.py:289-320,413-534 at H=4096: full-width decoder layers (fwd + dgrad), lm_head/CE at V=128256,
    full-width seg / depth / gen heads fwd + bwd + losses, projector and task-token gradients."""
    from visper_lm_amd.config import llama3_8b
    cfg = llama3_8b(num_hidden_layers=2, vit_layers=4)
    cfg.image_seg = dict(cfg.image_seg, seg_layer_indices="1")
    cfg.image_depth = dict(cfg.image_depth, depth_layer_indices="2")
    cfg.image_gen = dict(cfg.image_gen, img_layer_indices="2")
    got, _ = _run(cfg, 2, 128, "fullwidth_llama_L2", TIGHT)
    assert got["S"] == 127 + 576 + 24


def test_fold_norm_step_matches_unfolded_step(monkeypatch):
    """DESIGN section 2, deviation (v): which rounding points the decoder's RMSNorms take depends on the SHAPE — the fold (gamma in the
    derived weights, 1/rms as a row scale of the QKV / SwiGLU epilogues, sums of squares out of the residual GEMMs) needs tokens % 256 == 0 and
    one-wave-per-SIMD GEMM shapes, every other batch takes HF's rounding points.  Both must describe the same step: full-width Llama layers at a
    shape that folds (B = 2, S = 768 -> 1536 tokens), once with the fold and once with VP_FOLD_NORM=0, same weights and batch: losses, layer losses,
    hidden states and every gradient agree to bf16-rounding level (the other full-width tests of this file have S = 727: un-folded by shape)."""
    from visper_lm_amd.config import llama3_8b
    cfg = llama3_8b(num_hidden_layers=2, vit_layers=4)
    cfg.image_seg = dict(cfg.image_seg, seg_layer_indices="1")
    cfg.image_depth = dict(cfg.image_depth, depth_layer_indices="2")
    cfg.image_gen = dict(cfg.image_gen, img_layer_indices="2")
    from visper_lm_amd.engine import Engine
    W, batch = _weights(cfg), _batch(cfg, 2, 169)

    def run(fold):
        monkeypatch.setenv("VP_FOLD_NORM", "1" if fold else "0")
        eng = Engine(cfg)
        eng.load_weights(W)
        out = eng.train_step(_gpu(batch))
        torch.cuda.synchronize()
        assert out["plan"]["S"] == 768 and eng.last_fold == fold, (out["plan"]["S"], eng.last_fold)
        res = dict(loss=float(out["loss"]), text_loss=float(out["text_loss"]), hidden=out["hidden"].float().cpu(),
                   layer_losses={k: v.float().cpu() for k, v in out["layer_losses"].items()},
                   grads={k: eng.ps.g(k).detach().float().cpu().clone() for k in eng.ps.index})
        del eng, out
        torch.cuda.empty_cache()
        return res
    a, b = run(True), run(False)
    check("fold_vs_unfolded/loss_rel", rel(a["loss"], b["loss"]), 1e-3)
    check("fold_vs_unfolded/text_loss_rel", rel(a["text_loss"], b["text_loss"]), 1e-3)
    for key, trip in b["layer_losses"].items():
        for j, nm in enumerate(("emb", "sl1", "con")):
            if abs(float(trip[j])) > 1e-9:
                check(f"fold_vs_unfolded/layer_loss/{key[0]}@{key[1]}/{nm}_rel", rel(a["layer_losses"][key][j], trip[j]), 5e-3)
    check("fold_vs_unfolded/hidden_frob", _frob(a["hidden"], b["hidden"]), 2e-2)
    # gradients: the depth head (ReLU gates + LayerNorm backward) amplifies ANY bf16 rounding difference of its input state — the reference-style
    # bf16 CPU path sits 4.6e-2 from fp32 truth on the same parameters (TIGHT above) — so it gets that yardstick; everything else is tight
    worst = {"depth_head": (0.0, ""), "rest": (0.0, "")}
    for k, gb in b["grads"].items():
        if gb.numel() > 1 and float(gb.abs().max()) > 0:
            c, _ = grad_err(a["grads"][k], gb)
            grp = "depth_head" if "depth" in k else "rest"
            if c > worst[grp][0]:
                worst[grp] = (c, k)
    print("[parity] fold_vs_unfolded worst gradients:", worst)
    check("fold_vs_unfolded/worst_grad_one_minus_cos/rest", worst["rest"][0], 5e-3)
    check("fold_vs_unfolded/worst_grad_one_minus_cos/depth_head", worst["depth_head"][0], 7e-2)


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_config0_full_depth_vs_fp32_truth_and_reference_style_bf16_cpu_path():
    """BASELINE.json configs[0] exactly: random-init CLIP-ViT-L + Llama-3-8B (all 32 layers), 1 distill layer (seg@18), text length 128,
    2 images.  Three runs on identical weights / batch: the HIP step, the oracle in fp32 arithmetic (the numerical truth for these bf16
    weights) and the oracle executed like the reference's CPU PyTorch path (bf16 weights and activations, PyTorch bf16 ops).  Asserted:
    HIP vs truth and HIP vs the bf16 CPU path; logged beside them: the bf16 CPU path's own distance from the truth (the reference's
    noise floor; SURVEY 6 measured 1.6e-3 between two of its own runs)."""
    from visper_lm_amd.config import llama3_8b
    cfg = llama3_8b(aux_mode="seg")
    got, Wc, batch, tr = _hip_step(cfg, 2, 128)
    assert got["S"] == 127 + 576 + 8
    refb = _oracle_step(cfg, Wc, batch, tr, BF, got["rows"])
    print(f"[parity] config0: bf16 CPU oracle fwd+bwd {refb['seconds']:.1f} s")
    # measured (r02): HIP vs truth: loss 6.5e-5, seg loss 8e-6, hidden / logits 6.6e-2 (32 bf16 layers; the bf16 CPU path: 6.8e-2),
    # gradients 1 - cos <= 2.7e-3 (bf16 CPU path: 3.0e-3).  HIP vs the bf16 CPU path: loss 1.9e-4, seg contrastive term 1.9e-3.
    # round 6: the element-wise bounds of the 32-layer states were 0.2 / 0.3 for a measured 6.6e-2 / 7.9e-2.  Against fp32
    # truth they are now min(0.12, 1.5 x the bf16 CPU path's own deviation from that truth) (measured r05: 6.8e-2 / 8.2e-2 -> bounds 0.10 / 0.12);
    # HIP against the bf16 CPU path itself (two independent bf16 roundings of the same 32 layers; measured 7.7e-2 / 9.8e-2): 0.12 / 0.15
    deep = dict(TIGHT, inputs_embeds_frob=4e-2, inputs_embeds_max=4e-2, hidden_frob=0.12, logits_frob=0.12, hidden_max=0.15, logits_max=0.15,
                layer_loss=1e-2, grad_cos=1e-2, grad_norm=3e-2, grad_scalar_rel=0.3, grad_scalar_abs=3e-6)
    _compare("config0_vs_bf16_cpu_path", got, refb, deep)
    if _mem_available_gb() < 70:
        pytest.skip("fp32 truth leg needs ~40 GB of host memory")
    ref32 = _oracle_step(cfg, Wc, batch, tr, torch.float32, got["rows"])
    print(f"[parity] config0: fp32 CPU oracle fwd+bwd {ref32['seconds']:.1f} s")
    base = _compare("config0_INFO_bf16_cpu_path_vs_fp32_truth", refb, ref32, {})
    _compare("config0_vs_fp32_truth", got, ref32, dict(deep, state_vs_baseline=(1.5, 0.12)), baseline=base)


def test_fullwidth_phi3_long_context_vs_fp32_oracle():
    """configs[4] shapes: Phi-3-mini width, S=4096 (> sliding window 2047, so the window mask is live), D=96 attention, fused qkv_proj /
    gate_up_proj; two layers, heads at full width (depth head D=3072).  B=2 so that the contrastive term is live."""
    from visper_lm_amd.config import phi3_mini
    cfg = phi3_mini(num_hidden_layers=2, vit_layers=4)
    cfg.image_seg = dict(cfg.image_seg, seg_layer_indices="1")
    cfg.image_depth = dict(cfg.image_depth, depth_layer_indices="2")
    cfg.image_gen = dict(cfg.image_gen, img_layer_indices="2")
    got, _ = _run(cfg, 2, 3497, "fullwidth_phi3_S4096_L2", TIGHT)
    assert got["S"] == 4096
