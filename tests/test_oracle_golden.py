"""Pin the CPU oracle (oracle/visper_oracle.py) to golden vectors produced by the reference itself
(oracle/gen_golden.py).  fp32, CPU only."""
import json

import numpy as np
import pytest
import torch

from oracle import cases, weights as WT, visper_oracle as O


def _close(a, b, rtol=2e-4, atol=2e-5):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert np.allclose(a, b, rtol=rtol, atol=atol), f"max abs err {err}, ref max {np.abs(b).max()}"


@pytest.mark.parametrize("name,shp", [("gen", (3, 1, 1024)), ("depth", (3, 40, 1024)), ("seg", (3, 96, 6, 6))])
def test_emb_loss_units(name, shp):
    g = cases.load_golden("units.npz")
    p = WT.tensor(f"unit_pred_{name}", shp, 1.3).requires_grad_(True)
    t = WT.tensor(f"unit_tgt_{name}", shp, 1.0)
    s = torch.tensor(2.0, requires_grad=True)
    e, l1, c = O.emb_loss(p, torch.tensor([1.0, 0.0, 1.0]), t, s, 0.3)
    e.backward()
    _close([e.item(), l1.item(), c.item()], g[f"{name}_out"], 1e-5, 1e-7)
    _close(p.grad.numpy(), g[f"{name}_dpred"], 1e-4, 1e-9)
    _close(s.grad.item(), g[f"{name}_dscale"], 1e-4, 1e-8)
    _close(O.contrastive_loss(p.detach(), t, s.detach()).numpy(), g[f"{name}_con"], 1e-5, 1e-6)


def test_contrastive_saturation_and_nocontrastive():
    g = cases.load_golden("units.npz")
    p = WT.tensor("unit_pred_sat", (4, 8, 16), 1.0)
    t = WT.tensor("unit_tgt_sat", (4, 8, 16), 1.0)
    _close(O.contrastive_loss(p, t, torch.tensor(5.0)).numpy(), g["sat_con"], 1e-5, 1e-6)
    e, l1, c = O.emb_loss(p, torch.ones(4), t, None, 0.3)
    _close([float(e), float(l1), float(c)], g["nocon_out"], 1e-6, 1e-8)


@pytest.mark.parametrize("name,dims", [("rs_gen", (64, 1, 48, 64, 8)), ("rs_tile", (32, 16, 48, 40, 8)),
                                       ("rs_same", (32, 12, 48, 40, 12)), ("rs_mean", (32, 6, 48, 40, 4)),
                                       ("rs_deep", (32, 12, 48, 40, 12))])
def test_task_token_resampler(name, dims):
    g = cases.load_golden("units.npz")
    dim, nq, emb, out_dim, nlat = dims
    man = json.loads(str(g[f"{name}_manifest"]))
    W = {f"h.{k}": WT.param(f"{name}.{k}", s) for k, s in man.items()}
    x = WT.tensor(f"{name}.x", (2, 50, emb))
    lat = WT.tensor(f"{name}.lat", (2, nlat, emb))
    out = O.task_token_resampler(x, lat, W, "h.", dict(num_tokens=nq, num_heads=4, dim_head=32, depth=2 if name == "rs_deep" else 1))
    _close(out.numpy(), g[f"{name}_out"], 1e-4, 1e-5)


@pytest.fixture(scope="module")
def tiny():
    cfg, W, batch, g = cases.tiny_llama_case()
    tr = json.loads(str(g["trainable"]))
    for k in tr:
        W[k] = W[k].clone().requires_grad_(True)
    out = O.forward(W, batch, cfg)
    out["loss"].backward()
    return cfg, W, batch, g, out, tr


def test_e2e_manifest_names(tiny):
    cfg, W, batch, g, out, tr = tiny
    man = json.loads(str(g["manifest"]))
    # every parameter the oracle touches exists in the reference state dict with the same shape
    for k, v in W.items():
        assert tuple(man[k]) == tuple(v.shape), k
    assert int(g["n_hidden_states"]) == cfg.num_hidden_layers + 1


def test_e2e_forward_matches_reference(tiny):
    cfg, W, batch, g, out, tr = tiny
    _close(out["loss"].item(), g["keep_loss"], 1e-5, 1e-6)
    lg = out["logits"]
    assert tuple(lg.shape) == tuple(g["logits_shape"])
    _close(lg[:, ::41, ::997].detach().numpy(), g["logits_sub"], 1e-3, 2e-5)
    _close(torch.logsumexp(lg, -1)[:, ::7].detach().numpy(), g["logits_lse_sub"], 1e-5, 1e-5)
    hs = [out["inputs_embeds"]] + out["layer_states"]
    for li in (0, 2, 3, 4):
        _close(hs[li][:, ::13, ::3].detach().numpy(), g[f"hidden{li}_sub"], 1e-3, 2e-5)
    ll = g["keep_layer_losses"]
    shapes = json.loads(str(g["keep_layer_shapes"]))
    # reference call order: depth layers, seg layers, gen layers (ola_llama.py:139-141)
    mine = [out["layer_losses"][("depth", 2)], out["layer_losses"][("seg", 1)], out["layer_losses"][("seg", 2)],
            out["layer_losses"][("gen", 3)]]
    assert [len(s) for s in shapes] == [3, 4, 4, 3]
    for i, trip in enumerate(mine):
        _close([float(x) for x in trip], ll[i], 2e-5, 1e-6)
    _close(cases.sub(out["seg_embs"][0], 2048), g["seg_emb_sub"], 1e-3, 2e-5)
    _close(cases.sub(out["gen_embs"][0], 1024), g["gen_emb_sub"], 1e-3, 2e-5)
    assert int(g["depth_embs_len"]) == len(out["depth_embs"][0]) == 4


def test_e2e_dpt_depth_pred_matches_reference(tiny, tiny_phi3):
    """a11: the frozen DPT decoder (da_v2_head.py:260-321) + min-max normalisation (base_ola_vlm.py:462-470)."""
    for cfg, W, batch, g, out, tr in (tiny, tiny_phi3):
        dp = out["depth_preds"][0]
        assert tuple(dp.shape) == tuple(g["depth_preds_shape"]) == (2, 336, 336)
        assert float(dp.min()) == 0.0 and abs(float(dp.max()) - 1.0) < 1e-6
        _close(dp[:, ::5, ::5].numpy(), g["depth_pred_sub"], 2e-3, 2e-4)
        _close(float(dp.double().mean()), g["depth_pred_mean"], 1e-3, 1e-5)


def test_e2e_gradients_match_reference(tiny):
    cfg, W, batch, g, out, tr = tiny
    none_ref = set(json.loads(str(g["keep_grad_none"])))
    for k in tr:
        gr = W[k].grad
        if k in none_ref:
            assert gr is None or float(gr.abs().sum()) == 0.0, k
            continue
        assert gr is not None, k
        ref_norm = float(g[f"keep_gradnorm::{k}"])
        assert abs(float(gr.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, (k, float(gr.double().norm()), ref_norm)
        _close(cases.sub(gr, 256), g[f"keep_gradsub::{k}"], 2e-3, 1e-4 * ref_norm / max(1.0, gr.numel() ** 0.5) + 1e-9)


def test_e2e_as_released_mask_zeroing(tiny):
    cfg, W, batch, g, _, tr = tiny
    import copy
    cfg2 = copy.copy(cfg); cfg2.zero_masks = True
    W2 = {k: v.detach().clone().requires_grad_(k in tr) for k, v in W.items()}
    out = O.forward(W2, batch, cfg2)
    out["loss"].backward()
    _close(out["loss"].item(), g["released_loss"], 1e-5, 1e-6)
    _close(out["loss"].item(), out["text_loss"].item(), 0, 0)
    assert np.all(g["released_layer_losses"] == 0.0)
    for k in tr:
        if "_heads." in k or "logit_scale" in k:
            assert W2[k].grad is None or float(W2[k].grad.abs().sum()) == 0.0, k


# ------------------------------------------------------------------------------------------------ Phi-3 (config 5 path)
@pytest.fixture(scope="module")
def tiny_phi3():
    cfg, W, batch, g = cases.tiny_llama_case("phi3")
    tr = json.loads(str(g["trainable"]))
    for k in tr:
        W[k] = W[k].clone().requires_grad_(True)
    out = O.forward(W, batch, cfg)
    out["loss"].backward()
    return cfg, W, batch, g, out, tr


def test_phi3_e2e_forward_and_grads_match_reference(tiny_phi3):
    """OlaLlavaPhi3ForCausalLM: fused qkv_proj / gate_up_proj, NUM_SYS_TOKENS = 13, sliding window 300 < S = 653."""
    cfg, W, batch, g, out, tr = tiny_phi3
    assert O.num_sys_tokens(cfg) == 13 and cfg.sliding_window == 300
    _close(out["loss"].item(), g["keep_loss"], 1e-5, 1e-6)
    lg = out["logits"]
    assert tuple(lg.shape) == tuple(g["logits_shape"])
    _close(lg[:, ::41, ::997].detach().numpy(), g["logits_sub"], 1e-3, 2e-5)
    mine = [out["layer_losses"][("depth", 2)], out["layer_losses"][("seg", 1)], out["layer_losses"][("seg", 2)], out["layer_losses"][("gen", 3)]]
    for i, trip in enumerate(mine):
        _close([float(x.detach()) for x in trip], g["keep_layer_losses"][i], 2e-5, 1e-6)
    none_ref = set(json.loads(str(g["keep_grad_none"])))
    for k in tr:
        if k in none_ref:
            continue
        ref_norm = float(g[f"keep_gradnorm::{k}"])
        assert abs(float(W[k].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, k


def test_ift_stage_matches_reference_llava_llama():
    """NTP-only step of the reference's LlavaLlamaForCausalLM (llava_llama.py, llava_arch.py) with the whole LLM
    trainable — loss, logits and EVERY parameter gradient of the oracle against the reference's own autograd."""
    cfg, W, batch, g = cases.tiny_ift_case()
    tr = json.loads(str(g["trainable"]))
    W = {k: (v.clone().requires_grad_(True) if k in tr else v) for k, v in W.items()}
    out = O.forward(W, batch, cfg)
    out["loss"].backward()
    _close(out["loss"].item(), g["loss"], 1e-5, 1e-6)
    assert tuple(out["logits"].shape) == tuple(g["logits_shape"])
    _close(out["logits"][:, ::41, ::997].detach().numpy(), g["logits_sub"], 1e-3, 2e-5)
    assert len(tr) == 43 and "lm_head.weight" in tr and "model.embed_tokens.weight" in tr
    for k in tr:
        ref_norm = float(g[f"gradnorm::{k}"])
        assert abs(float(W[k].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, k
        _close(cases.sub(W[k].grad, 128), g[f"gradsub::{k}"], 2e-3, 1e-7 + 1e-4 * float(np.abs(g[f"gradsub::{k}"]).max()))


def test_ift_stage_with_pt_task_tokens_matches_reference_llava_llama():
    """VERDICT r2 missing-2: the reference's own PT -> IFT hand-off.  LlavaLlamaForCausalLM built from a PT-stage config
    (num_task_tokens 8, task_token_format "emb") splices the RAW (576, H) depth / seg parameters and the 8 gen rows behind the image
    (llava_arch.py:250-293): S = 58 + 576 + 576 + 576 + 8.  Loss, logits and every parameter gradient incl. the three token tensors."""
    cfg, W, batch, g = cases.tiny_ift_tok_case()
    tr = json.loads(str(g["trainable"]))
    W = {k: (v.clone().requires_grad_(True) if k in tr else v) for k, v in W.items()}
    out = O.forward(W, batch, cfg)
    out["loss"].backward()
    assert tuple(out["logits"].shape) == tuple(g["logits_shape"]) and out["logits"].shape[1] == 58 + 576 * 3 + 8
    _close(out["loss"].item(), g["loss"], 1e-5, 1e-6)
    _close(out["logits"][:, ::41, ::997].detach().numpy(), g["logits_sub"], 1e-3, 2e-5)
    assert len(tr) == 46 and all(f"model.special_{t}_tokens" in tr for t in ("depth", "seg", "gen"))
    for k in tr:
        ref_norm = float(g[f"gradnorm::{k}"])
        assert abs(float(W[k].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, k
        _close(cases.sub(W[k].grad, 128), g[f"gradsub::{k}"], 2e-3, 1e-7 + 1e-4 * float(np.abs(g[f"gradsub::{k}"]).max()))
    # "text": the reference calls embed_tokens on the float parameters -> F.embedding raises; recorded by the generator
    assert str(g["text_format_error"]).startswith("RuntimeError")


def test_pt_step_without_task_tokens_matches_reference():
    """VERDICT r2 missing-5: num_task_tokens == 0 -> GenHead / DepthHead / OneFormerSegHead around the plain Resampler with its own
    latents (base_ola_vlm.py:120-169, 429-430; resampler.py:120-165), whole layer state as head input."""
    cfg, W, batch, g = cases.tiny_nt0_case()
    tr = json.loads(str(g["trainable"]))
    assert not any("special_" in k for k in W) and sum(k.endswith("projector.latents") for k in tr) == 4
    W = {k: (v.clone().requires_grad_(True) if k in tr else v) for k, v in W.items()}
    out = O.forward(W, batch, cfg)
    out["loss"].backward()
    assert tuple(out["logits"].shape) == tuple(g["logits_shape"]) and out["logits"].shape[1] == 58 + 576
    _close(out["loss"].item(), g["loss"], 1e-5, 1e-6)
    mine = [out["layer_losses"][("depth", 2)], out["layer_losses"][("seg", 1)], out["layer_losses"][("seg", 2)], out["layer_losses"][("gen", 3)]]
    for i, trip in enumerate(mine):
        _close([float(x.detach()) for x in trip], g["layer_losses"][i], 2e-5, 1e-6)
    _close(cases.sub(out["seg_embs"][0], 2048), g["seg_emb_sub"], 1e-3, 2e-5)
    none_ref = set(json.loads(str(g["grad_none"])))
    for k in tr:
        if k in none_ref:
            continue
        ref_norm = float(g[f"gradnorm::{k}"])
        assert abs(float(W[k].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, k


def test_pt_step_without_intermediate_depth_matches_reference():
    """VERDICT r5 missing-2: image_depth["use_intermediate_depth"] = False (base_ola_vlm.py:132,462-466; da_v2_head.py:437-455): no linear_1..3
    parameters, the loss compares visual_feats itself, depth_embs entries have ONE feature map and the DPT decoder runs on [feats[0]] * 4."""
    cfg, W, batch, g = cases.tiny_noid_case()
    tr = json.loads(str(g["trainable"]))
    assert not any(".linear_" in k for k in W) and cfg.image_depth["use_intermediate_depth"] is False
    W = {k: (v.clone().requires_grad_(True) if k in tr else v) for k, v in W.items()}
    out = O.forward(W, batch, cfg)
    out["loss"].backward()
    _close(out["loss"].item(), g["loss"], 1e-5, 1e-6)
    assert json.loads(str(g["layer_shapes"]))[0] == [2, 576, 1024] and int(g["depth_embs_len"]) == 1 == len(out["depth_embs"][0])
    mine = [out["layer_losses"][("depth", 2)], out["layer_losses"][("seg", 1)], out["layer_losses"][("seg", 2)], out["layer_losses"][("gen", 3)]]
    for i, trip in enumerate(mine):
        _close([float(x.detach()) for x in trip], g["layer_losses"][i], 2e-5, 1e-6)
    _close(cases.sub(out["depth_embs"][0][0], 2048), g["depth_emb_sub"], 1e-3, 2e-5)
    dp = out["depth_preds"][0]
    assert tuple(dp.shape) == tuple(g["depth_preds_shape"])
    assert float(np.abs(dp[:, ::5, ::5].detach().float().numpy() - g["depth_pred_sub"]).max()) < 2e-4
    none_ref = set(json.loads(str(g["grad_none"])))
    for k in tr:
        if k in none_ref:
            continue
        ref_norm = float(g[f"gradnorm::{k}"])
        assert abs(float(W[k].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, k


@pytest.mark.parametrize("name,dims", [("rs_plain", (32, 12, 48, 40, 1)), ("rs_plain_deep", (64, 5, 48, 24, 2))])
def test_plain_resampler(name, dims):
    g = cases.load_golden("units.npz")
    dim, nq, emb, out_dim, depth = dims
    man = json.loads(str(g[f"{name}_manifest"]))
    W = {f"h.{k}": WT.param(f"{name}.{k}", s) for k, s in man.items()}
    assert tuple(man["latents"]) == (1, nq, dim)
    out = O.resampler(WT.tensor(f"{name}.x", (2, 50, emb)), W, "h.", dict(num_tokens=nq, num_heads=4, dim_head=32, depth=depth))
    _close(out.numpy(), g[f"{name}_out"], 1e-4, 1e-5)


@pytest.mark.parametrize("name,shp", [("rep_gen", (1, 1024)), ("rep_depth", (40, 256))])
def test_emb_loss_batch_repeat_branch(name, shp):
    """base_ola_vlm.py:292-299: 4 predictions against 2 targets -> targets.repeat(2, 1, 1), mask.repeat(2, 1, 1)."""
    g = cases.load_golden("units.npz")
    p = WT.tensor(f"unit_pred_{name}", (4, *shp), 1.3).requires_grad_(True)
    t = WT.tensor(f"unit_tgt_{name}", (2, *shp), 1.0)
    s = torch.tensor(2.0, requires_grad=True)
    e, l1, c = O.emb_loss(p, torch.tensor([1.0, 0.5]), t, s, 0.3)
    e.backward()
    _close([e.item(), l1.item(), c.item()], g[f"{name}_out"], 1e-5, 1e-7)
    _close(p.grad.numpy(), g[f"{name}_dpred"], 1e-4, 1e-9)
    _close(s.grad.item(), g[f"{name}_dscale"], 1e-4, 1e-8)
    assert str(g["rep_rank4_error"]).startswith("RuntimeError")          # the reference's 3-argument repeat on a rank-4 (seg) target
    with pytest.raises(RuntimeError):
        O.emb_loss(WT.tensor("unit_pred_rep4", (4, 8, 3, 3)), torch.ones(2), WT.tensor("unit_tgt_rep4", (2, 8, 3, 3)), None, 0.3)


def test_dinov2_depth_teacher_matches_reference():
    """the DINOv2 depth-teacher target (mean of 4 normed intermediate patch-token maps; base_ola_vlm.py:347-365 ->
    depth_anything_v2/dinov2.py) against the reference's own DinoVisionTransformer, incl. the bicubic 37x37 -> 24x24 position grid."""
    from oracle import weights as WT
    g = cases.load_golden("dinov2_teacher.npz")
    man = json.loads(str(g["manifest"]))
    W = cases.dinov2_weights(man)
    dims = json.loads(str(g["dims"]))
    images = WT.tensor("dino_images", (2, 3, 336, 336))
    with torch.no_grad():
        tgt = O.dinov2_depth_target(images, W, dims["num_heads"], [int(t) for t in g["taps"]])
    assert tuple(tgt.shape) == tuple(g["target_shape"])
    _close(tgt[:, ::7, ::3].numpy(), g["target_sub"], 1e-3, 2e-5)
    _close(float(tgt.double().mean()), g["target_mean"], 1e-4, 1e-6)
    _close(float(tgt.double().std()), g["target_std"], 1e-4, 1e-6)


def test_clip_image_embed_teacher_matches_hf():
    """the generation teacher target `pipe.image_encoder(x).image_embeds` (base_ola_vlm.py:323-332) against HF's own
    CLIPVisionModelWithProjection (tests/golden/clip_embed_teacher.npz)."""
    from oracle import weights as WT
    g = cases.load_golden("clip_embed_teacher.npz")
    dims = json.loads(str(g["dims"]))
    W = {k: WT.param(k, s) for k, s in json.loads(str(g["manifest"])).items()}
    images = WT.tensor("clip_embed_images", (2, 3, 224, 224))
    with torch.no_grad():
        emb = O.clip_image_embeds(images, W, dims["num_attention_heads"], dims["patch_size"], act=dims["hidden_act"])
    assert tuple(emb.shape) == tuple(g["embeds"].shape)
    _close(emb.numpy(), g["embeds"], 1e-3, 2e-5)


def test_swin_seg_teacher_matches_hf():
    """the segmentation teacher target (Swin backbone last feature map -> 24 x 24; base_ola_vlm.py:382-397,
    oneformer_head.py:11-69) against HF's own SwinBackbone: W-MSA / SW-MSA with relative position bias and cyclic-shift masks, patch merging."""
    from oracle import weights as WT
    g = cases.load_golden("swin_teacher.npz")
    dims = json.loads(str(g["dims"]))
    W = cases.swin_weights(json.loads(str(g["manifest"])))
    images = WT.tensor("swin_images", (2, 3, 384, 384))
    with torch.no_grad():
        tgt = O.swin_seg_target(images, W, dims["depths"], dims["num_heads"], window=dims["window_size"], patch=dims["patch_size"])
    assert tuple(tgt.shape) == tuple(g["target_shape"])
    _close(tgt[:, ::5, ::3, ::3].numpy(), g["target_sub"], 1e-3, 5e-5)
    _close(float(tgt.double().std()), g["target_std"], 1e-4, 1e-6)


@pytest.mark.parametrize("tag", ["short", "noimg"])
def test_padded_rows_match_reference(tag):
    """tests/golden/tiny_llama_ragged.npz (oracle/gen_golden.py run_tiny_ragged): a right-padded batch through the REFERENCE.  Padded query rows
    are never masked; with no position_ids from the caller the reference hands `position_ids=None` on (ola_arch.py:439-440) and HF numbers all rows
    of the padded tensor 0..S-1, and forward_emb_predictor feeds those rows to the shorter sample's heads (base_ola_vlm.py:413-443).  The oracle
    must reproduce the hidden states of EVERY row of the short sample, every layer's loss triple and the per-sample predictions."""
    cfg, W, batch, _ = cases.tiny_llama_case()
    g = cases.load_golden("tiny_llama_ragged.npz")
    batch = dict(batch, input_ids=torch.from_numpy(g[f"{tag}_input_ids"]), attention_mask=torch.from_numpy(g[f"{tag}_attention_mask"]),
                 labels=torch.from_numpy(g[f"{tag}_labels"]))
    with torch.no_grad():
        out = O.forward(W, batch, cfg, need_logits=False)
    _close(float(out["loss"]), g[f"{tag}_loss"], 2e-5, 1e-6)
    hs = out["layer_states"]
    _close(hs[-1][1, :, ::3].numpy(), g[f"{tag}_hidden_last_sample1"], 1e-3, 2e-4)          # post-norm final state: real AND padded rows
    _close(hs[1][1, :, ::3].numpy(), g[f"{tag}_hidden2_sample1"], 1e-3, 2e-4)               # hidden_states[2] = output of layer 2
    shapes = json.loads(str(g[f"{tag}_layer_shapes"]))
    names = [("depth", 2), ("seg", 1), ("seg", 2), ("gen", 3)]
    assert len(shapes) == len(names)
    for i, key in enumerate(names):
        _close([float(x) for x in out["layer_losses"][key]], g[f"{tag}_layer_losses"][i], 2e-5, 1e-6)
