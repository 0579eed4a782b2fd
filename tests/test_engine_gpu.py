"""End-to-end GPU parity of the PT train step (C-ABI kernels composed by Engine) against the CPU oracle and
the golden vectors produced by the reference itself.  bf16 compute vs fp32 golden: tolerances are stated."""
import copy
import os
import json

import numpy as np
import pytest
import torch

from parity import check, log, max_rel, grad_err, scalar_grad_yardstick, scalar_grad_bound

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _trip_err(mine, theirs):
    """worst relative error over the (emb, sl1, con) triple of one head-layer call; absolute 1e-6 floor for exact zeros."""
    return max(abs(float(a) - float(b)) / max(abs(float(b)), 1e-6) for a, b in zip(mine, theirs))


def _bf16_cpu_layer_dev(O, ocfg, W, batch, ref32):
    """The reference-style CPU path — bf16 weights, bf16 activations, PyTorch's bf16 CPU ops (what north_star compares against) — measured against
    the fp32-math oracle on the same bf16-rounded weights: {head-layer key: worst relative deviation of its (emb, sl1, con) triple}.  This is the
    dtype's own noise floor; a HIP layer loss is held to max(1e-3 (north_star), 1.5 x this)."""
    Wb = {k: v.detach().to(BF) for k, v in W.items()}
    bb = {k: (v.to(BF) if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    with torch.no_grad():
        refb = O.forward(Wb, bb, ocfg, need_logits=False)
    return {key: _trip_err([float(x) for x in refb["layer_losses"][key]], [float(x) for x in trip]) for key, trip in ref32["layer_losses"].items()}


LAYER_LOSS_CAP = 1.2e-2       # the fixed bound of rounds 1-3: a noisy CPU bf16 run can never loosen the bar past it


def _layer_loss_bound(dev):
    return min(LAYER_LOSS_CAP, max(1e-3, 1.5 * dev))


def _to_gpu_batch(batch):
    return {k: (v.cuda() if (k == "images" or k.endswith("_target") or k.endswith("_mask")) else v) for k, v in batch.items()}


@pytest.fixture(scope="module")
def tiny():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_llama_case()
    cfg = VisperConfig(**vars(ocfg), depth_decoder=True)      # a11: also run the frozen DPT decoder (depth_preds)
    eng = Engine(cfg)
    eng.load_weights(W)
    eng.keep_logits = True
    out = eng.train_step(_to_gpu_batch(batch))
    torch.cuda.synchronize()
    grads = {k: eng.ps.g(k).detach().float().cpu().clone() for k in eng.ps.index}
    # oracle with bf16-rounded weights/inputs but fp32 arithmetic (isolates kernel error from quantisation of inputs)
    tr = json.loads(str(g["trainable"]))
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    cpu_dev = _bf16_cpu_layer_dev(O, ocfg, W, batch, ref)
    return dict(cfg=cfg, ocfg=ocfg, W=W, Wq=Wq, batch=batch, g=g, out=out, grads=grads, ref=ref, tr=tr, eng=eng, cpu_dev=cpu_dev)


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)


def test_dpt_depth_pred_matches_oracle_and_reference_golden(tiny):
    """a11 (da_v2_head.py:260-321 + base_ola_vlm.py:462-470): ~25 bf16 convolutions deep and min-max normalised to [0, 1], so
    the tolerance is stated on the map itself: mean |err| < 1e-2, max |err| < 8e-2 against the fp32 oracle on the SAME bf16
    features, and mean |err| < 2e-2 against the reference's own fp32 output (which also differs in the upstream features)."""
    from oracle import visper_oracle as O
    eng, out, g, Wq = tiny["eng"], tiny["out"], tiny["g"], tiny["Wq"]
    dp = out["depth_preds"][0].float().cpu()
    assert tuple(dp.shape) == (2, 336, 336) and float(dp.min()) == 0.0 and abs(float(dp.max()) - 1.0) < 1e-2
    feats = [f.float().cpu() for f in out["depth_feats"][0]]
    with torch.no_grad():
        ref = O.dpt_depth_pred(feats, {k: v.detach() for k, v in Wq.items()})
    err = (dp - ref).abs()
    check("tiny/depth_pred_mean_abs_vs_oracle", err.mean(), 1e-2)
    check("tiny/depth_pred_max_abs_vs_oracle", err.max(), 8e-2)
    gerr = (dp[:, ::5, ::5].numpy() - g["depth_pred_sub"])
    check("tiny/depth_pred_mean_abs_vs_reference_golden", np.abs(gerr).mean(), 2e-2)


def test_losses_match_oracle_and_reference_golden(tiny):
    """north-star tolerance: losses within 1e-3 relative of the reference's CPU path.  Bounds = measured error x ~3 (parity.check logs
    the measured values): vs the fp32-math oracle on the same bf16 weights, and vs the reference's own fp32 golden (which also carries
    the bf16 rounding of the weights themselves)."""
    out, ref, g = tiny["out"], tiny["ref"], tiny["g"]
    check("tiny/text_loss_rel_vs_oracle", rel(out["text_loss"], ref["text_loss"]), 1e-3)
    check("tiny/loss_rel_vs_oracle", rel(out["loss"], ref["loss"]), 1e-3)
    check("tiny/loss_rel_vs_reference_golden", rel(out["loss"], g["keep_loss"]), 1e-3)
    names = [("depth", 2), ("seg", 1), ("seg", 2), ("gen", 3)]
    for i, key in enumerate(names):
        mine = out["layer_losses"][key].float().cpu().numpy()
        theirs = np.array([float(x) for x in ref["layer_losses"][key]])
        dev = tiny["cpu_dev"][key]                                       # the reference-style bf16 CPU path's own distance from fp32 truth
        log(f"tiny/layer_loss/{key[0]}@{key[1]}_INFO_bf16_cpu_path_vs_fp32_truth", dev)
        check(f"tiny/layer_loss/{key[0]}@{key[1]}_vs_oracle", _trip_err(mine, theirs), _layer_loss_bound(dev))
        check(f"tiny/layer_loss/{key[0]}@{key[1]}_vs_reference_golden", _trip_err(mine, g["keep_layer_losses"][i]), 1e-2)


def test_hidden_states_and_logits(tiny):
    out, ref = tiny["out"], tiny["ref"]
    check("tiny/inputs_embeds_maxrel", max_rel(out["inputs_embeds"].cpu(), ref["inputs_embeds"].detach()), 1.5e-2)
    check("tiny/hidden_maxrel", max_rel(out["hidden"].cpu(), ref["hidden"].detach()), 3e-2)
    lg, rl = out["logits"].float().cpu(), ref["logits"].detach()
    assert lg.shape == rl.shape
    check("tiny/logits_maxrel", max_rel(lg, rl), 3e-2)


def test_labelled_row_compaction_is_exact(tiny):
    """Only the rows that carry a label go through lm_head + cross-entropy + the d_hidden GEMM (with or without keep_logits; the label-less rows
    take a forward-only lm_head pass when the logits are wanted): same loss, bit-identical gradients AND bit-identical fp32 logits as sending
    every row through all three, literally as ola_llama.py:121-136 does (Engine.lm_head_all_rows)."""
    eng = tiny["eng"]
    batch = _to_gpu_batch(tiny["batch"])
    plan = eng.build_plan(tiny["batch"]["input_ids"], tiny["batch"]["attention_mask"], tiny["batch"]["labels"])
    assert 0 < plan["n_valid"] < plan["B"] * plan["S"]                      # the compacted path really runs
    eng.lm_head_all_rows = True
    try:
        full = eng.train_step(batch)
        torch.cuda.synchronize()
        gfull = {k: eng.ps.g(k).detach().float().cpu().clone() for k in eng.ps.index}
    finally:
        eng.lm_head_all_rows = False
    assert full["logits"].dtype == torch.float32 and tiny["out"]["logits"].dtype == torch.float32
    assert torch.equal(full["logits"], tiny["out"]["logits"])                # compacted + forward-only rows == all rows, bit for bit
    assert torch.equal(full["text_loss"], tiny["out"]["text_loss"]) or rel(full["text_loss"], tiny["out"]["text_loss"]) < 1e-6
    for k in eng.ps.index:
        assert torch.equal(gfull[k], tiny["grads"][k]), k
    eng.keep_logits = False
    try:
        out = eng.train_step(batch)
        torch.cuda.synchronize()
    finally:
        eng.keep_logits = True
    assert "logits" not in out
    check("tiny/compact/text_loss_rel", rel(out["text_loss"], tiny["out"]["text_loss"]), 1e-6)
    check("tiny/compact/loss_rel", rel(out["loss"], tiny["out"]["loss"]), 1e-6)
    for k in eng.ps.index:
        assert torch.equal(eng.ps.g(k).detach().float().cpu(), tiny["grads"][k]), k


def test_gradients_match_oracle(tiny):
    grads, Wq, g = tiny["grads"], tiny["Wq"], tiny["g"]
    none_ref = set(json.loads(str(g["keep_grad_none"])))
    for k in tiny["tr"]:
        mine = grads[k].reshape(-1)
        if k in none_ref:
            assert float(mine.abs().sum()) == 0.0, k        # unused params (depth linear_2/3): zero-filled
            continue
        theirs = Wq[k].grad.reshape(-1)
        if mine.numel() == 1:                                # logit scales: a sum of a few signed terms
            check(f"tiny/grad/{k}_abs", abs(float(mine) - float(theirs)), 0.05 * abs(float(theirs)) + 1e-3)
            continue
        c, n = grad_err(mine, theirs)
        check(f"tiny/grad/{k}/one_minus_cos", c, 1.5e-2)
        check(f"tiny/grad/{k}/norm_dev", n, 2e-2)


def test_as_released_mask_zeroing(tiny):
    """with the in-place mask.zero_() every embedding loss and head gradient is exactly 0."""
    from visper_lm_amd.engine import Engine
    cfg = copy.copy(tiny["cfg"])
    cfg.zero_masks = True
    eng = Engine(cfg)
    eng.load_weights(tiny["W"])
    out = eng.train_step(_to_gpu_batch(tiny["batch"]))
    check("tiny/released_loss_rel_vs_reference_golden", rel(out["loss"], tiny["g"]["released_loss"]), 1e-3)
    assert float(out["loss"]) == float(out["text_loss"])
    for k in eng.ps.index:
        if "_heads." in k or k.endswith("logit_scale"):
            assert float(eng.ps.g(k).abs().sum()) == 0.0, k


def test_side_stream_schedule_is_bit_identical(tiny, monkeypatch):
    """The heads run on a side stream (forked after the decoder forward, joined in the decoder backward) and, for a batch flagged
    `images_resident`, so does the frozen tower (no dependency on the current stream).  Scheduling only: loss, layer losses and every gradient are
    bitwise those of the serial order (VP_HEADS_STREAM=0 / VP_TOWER_STREAM=0), also over back-to-back steps with the host running ahead."""
    eng, gb = tiny["eng"], _to_gpu_batch(tiny["batch"])
    torch.cuda.synchronize()

    def run(resident, n=3):
        res = []
        for _ in range(n):                                   # no sync between steps: step t+1's tower may run under step t's backward
            out = eng.train_step({**gb, "images_resident": resident})
            res.append((out["loss"].clone(), {k: v.clone() for k, v in out["layer_losses"].items()}, eng.ps.grad.clone()))
        torch.cuda.synchronize()
        return res

    monkeypatch.setenv("VP_HEADS_STREAM", "0")
    monkeypatch.setenv("VP_TOWER_STREAM", "0")
    serial = run(False, 1)[0]
    monkeypatch.setenv("VP_HEADS_STREAM", "1")
    monkeypatch.setenv("VP_TOWER_STREAM", "1")
    for resident in (False, True):
        for loss, ll, grad in run(resident):
            assert torch.equal(loss, serial[0])
            assert all(torch.equal(ll[k], serial[1][k]) for k in serial[1])
            assert torch.equal(grad, serial[2])


def test_optimizer_step_reduces_loss(tiny):
    from visper_lm_amd.engine import Engine
    eng = Engine(tiny["cfg"])
    eng.load_weights(tiny["W"])
    b = _to_gpu_batch(tiny["batch"])
    l0 = float(eng.train_step(b)["loss"])
    for _ in range(3):
        eng.ps.adamw_step(lr=1e-3)
        l1 = float(eng.train_step(b)["loss"])
    assert l1 < l0, (l0, l1)


def test_smoke_entry():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__ as ge
    ge.smoke()


def test_phi3_path_matches_oracle_and_reference_golden():
    """BASELINE configs[4] path: OlaLlavaPhi3 (fused qkv/gate_up weights, NUM_SYS_TOKENS 13, sliding-window attention)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_llama_case("phi3")
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    check("tiny_phi3/loss_rel_vs_reference_golden", rel(out["loss"], g["keep_loss"]), 1e-3)
    names = [("depth", 2), ("seg", 1), ("seg", 2), ("gen", 3)]
    for i, key in enumerate(names):
        mine = out["layer_losses"][key].float().cpu().numpy()
        check(f"tiny_phi3/layer_loss/{key[0]}@{key[1]}_vs_reference_golden", _trip_err(mine, g["keep_layer_losses"][i]), 1e-2)
    none_ref = set(json.loads(str(g["keep_grad_none"])))
    # logit scales: |g| ~ 1e-6 (a sum of a few signed terms): absolute bound tied to what the bf16 CPU path itself deviates by
    scal = [k for k in eng.ps.index if eng.ps.g(k).numel() == 1 and k not in none_ref]
    yard = scalar_grad_yardstick(O, ocfg, W, batch, scal)
    for k in eng.ps.index:
        got = float(eng.ps.g(k).float().norm())
        ref = 0.0 if k in none_ref else float(g[f"keep_gradnorm::{k}"])
        if ref == 0.0:
            assert got == 0.0, k
        elif k in yard:
            log(f"tiny_phi3/gradnorm/{k}_INFO_bf16_cpu_path_abs_dev", yard[k][1])
            check(f"tiny_phi3/gradnorm/{k}_abs", abs(got - ref), scalar_grad_bound(ref, yard[k][1], 0.3))
        else:
            check(f"tiny_phi3/gradnorm/{k}_rel", abs(got - ref) / ref, 5e-2)


def test_convnext_tower_matches_oracle():
    """BASELINE configs[3] path (CLIP-ConvNeXt trunk as image encoder).  Oracle is UNPINNED for this tower (timm/open_clip
    absent): this is a self-consistency check HIP vs the CPU restatement of the public ConvNeXt definition."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import visper_oracle as O, weights as WT
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    from visper_lm_amd.params import param_shapes
    cfg = VisperConfig(mm_vision_tower="CLIP-convnext_tiny-res768", cnx_dims=(64, 64, 128, 192), cnx_depths=(1, 1, 2, 1),
                       vocab_size=1024, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4,
                       num_key_value_heads=2, aux_mode="", num_task_tokens=0)
    shapes = param_shapes(cfg)
    assert cfg.mm_hidden_size == 192 and "model.vision_tower.vision_tower.stages.2.blocks.1.gamma" in shapes
    W = {k: WT.param(k, s) for k, s in shapes.items()}
    for k in W:
        if k.endswith(".gamma"):
            W[k] = WT.tensor(k, shapes[k], 0.5)                      # layer scale large enough to matter
    images = WT.tensor("cnx_images", (1, 3, 768, 768))
    ocfg = O.make_config(**{k: v for k, v in cfg.to_dict().items() if k in vars(O.make_config())})
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    ref = O.convnext_features(images.to(BF).float(), Wq, ocfg)       # (1, 576, 192)
    eng = Engine(cfg)
    eng.load_weights(W)
    got = eng.vit_forward(images.cuda()).float().cpu().view(1, 576, 192)
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 3e-2, float(err)


def test_convnext_tower_matches_reference_golden():
    """Row a2 pin on the GPU: the HIP ConvNeXt tower against tests/golden/convnext.npz = the REFERENCE's own CLIPConvNextVisionTower._forward
    (clip_convnext_encoder.py:150-174) run over transformers.ConvNextModel's modules (oracle/gen_golden.py run_convnext).  The golden is fp32; the
    tower computes in bf16 — the oracle on the same bf16-rounded weights is logged beside it so the bound is dtype, not algorithm."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import numpy as np
    from oracle import visper_oracle as O, weights as WT
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    from visper_lm_amd.params import param_shapes
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "convnext.npz"))
    dims, depths, px = tuple(int(x) for x in g["dims"]), tuple(int(x) for x in g["depths"]), int(g["px"])
    cfg = VisperConfig(mm_vision_tower="CLIP-convnext-pin", cnx_dims=dims, cnx_depths=depths, cnx_eps=float(g["eps"]), cnx_image=px,
                       vocab_size=1024, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4,
                       num_key_value_heads=2, aux_mode="", num_task_tokens=0)
    shapes = param_shapes(cfg)
    W = {k: WT.param(k, s) for k, s in shapes.items()}
    for k in W:
        if k.endswith(".gamma"):
            W[k] = WT.tensor(k, shapes[k], 0.5)
    images = WT.tensor("cnx_pin_images", (2, 3, px, px))
    ref = torch.from_numpy(g["features"])
    ocfg = O.make_config(**{k: v for k, v in cfg.to_dict().items() if k in vars(O.make_config())})
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    orc = O.convnext_features(images.to(BF).float(), Wq, ocfg)
    eng = Engine(cfg)
    eng.load_weights(W)
    got = eng.vit_forward(images.cuda()).float().cpu().view(ref.shape)
    check("convnext_pin/oracle_bf16_weights_vs_reference_golden", max_rel(orc, ref), 2e-2)
    check("convnext_pin/hip_vs_reference_golden", max_rel(got, ref), 4e-2)
    check("convnext_pin/hip_vs_oracle_same_bf16_weights", max_rel(got, orc), 3e-2)
    cos = torch.nn.functional.cosine_similarity(got.reshape(1, -1), ref.reshape(1, -1)).item()
    check("convnext_pin/hip_vs_reference_golden_1-cos", 1.0 - cos, 1e-3)


def _edge_case(ocfg_kw, mutate, min_cos=0.985, tag="edge", max_norm=3e-2):
    """Engine vs the fp32 oracle (same bf16-rounded weights / inputs) on a mutated copy of the tiny Llama case."""
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_llama_case()
    ocfg = O.make_config(**{**vars(ocfg), **ocfg_kw})
    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    mutate(batch)
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    tr = json.loads(str(g["trainable"]))
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    check(f"{tag}/text_loss_rel", rel(out["text_loss"], ref["text_loss"]), 1e-3)
    check(f"{tag}/loss_rel", rel(out["loss"], ref["loss"]), 1e-3)
    cpu_dev = _bf16_cpu_layer_dev(O, ocfg, W, batch, ref)
    for key, trip in ref["layer_losses"].items():
        mine = out["layer_losses"][key].float().cpu().numpy()
        log(f"{tag}/layer_loss/{key[0]}@{key[1]}_INFO_bf16_cpu_path_vs_fp32_truth", cpu_dev[key])
        check(f"{tag}/layer_loss/{key[0]}@{key[1]}", _trip_err(mine, [float(x) for x in trip]), _layer_loss_bound(cpu_dev[key]))
    for k in eng.ps.index:
        got = eng.ps.g(k).detach().float().cpu()
        want = Wq[k].grad
        if want is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        mine, theirs = got.reshape(-1), want.reshape(-1)
        if mine.numel() == 1:                                      # logit scales: a sum of a few signed terms, so absolute slack too
            check(f"{tag}/grad/{k}_abs", abs(float(mine) - float(theirs)), 0.05 * abs(float(theirs)) + 1e-3)
            continue
        if float(theirs.norm()) == 0.0:
            assert float(mine.norm()) < 1e-6, k
            continue
        c, n = grad_err(mine, theirs)
        check(f"{tag}/grad/{k}/one_minus_cos", c, 1.0 - min_cos)
        check(f"{tag}/grad/{k}/norm_dev", n, max_norm)
    return out, ref


def test_edge_ragged_right_padded_batch():
    """ola_arch.py:337-338 strips padding by attention_mask, :408-427 right-pads the spliced sequences: sample 1 is 17 tokens
    shorter than sample 0, so its tail rows are padding (labels -100, keys masked by kv_len)."""
    def mutate(b):
        b["attention_mask"][1, 42:] = False
    out, ref = _edge_case({}, mutate, tag="edge_ragged")
    assert out["plan"]["S"] == ref["labels"].shape[1]


def test_edge_sample_without_image():
    """ola_arch.py:344-355: a sample with no <image> token consumes an empty feature slot and carries no image / task rows."""
    def mutate(b):
        b["input_ids"][1, 38] = 7
    # the text-only sample feeds ~600 rows of padding-position states into every head's cross-attention.  (Rounds 1-3 ran this case with a 0.90
    # cosine bar: the ORACLE numbered padded rows 0 instead of their row index — fixed in round 4 against tests/golden/tiny_llama_ragged.npz; measured
    # now: 1 - cos 4.3e-4, norm 5.6e-3, the default bars hold.)
    _edge_case({}, mutate, tag="edge_no_image")


@pytest.mark.parametrize("tag", ["short", "noimg"])
def test_padded_rows_match_reference_golden(tag):
    """tests/golden/tiny_llama_ragged.npz = the REFERENCE on a right-padded batch (sample 1 shorter / without an image).  Padded query rows are
    real rows in the reference (never masked, positions 0..S-1 over the padded tensor: ola_arch.py:439-440) and the shorter sample's heads read
    them (base_ola_vlm.py:413-443), so the engine must reproduce their hidden states, not just the real rows': final hidden state of EVERY row
    of sample 1, loss, and every layer's loss triple against the reference's own numbers (fp32 golden vs bf16 compute: bounds are the dtype's)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, _ = cases.tiny_llama_case()
    g = cases.load_golden("tiny_llama_ragged.npz")
    batch = dict(batch, input_ids=torch.from_numpy(g[f"{tag}_input_ids"]), attention_mask=torch.from_numpy(g[f"{tag}_attention_mask"]),
                 labels=torch.from_numpy(g[f"{tag}_labels"]))
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    n = int(out["plan"]["lens_host"][1])
    hid = out["hidden"].float().cpu()[1, :, ::3]
    ref = torch.from_numpy(g[f"{tag}_hidden_last_sample1"])
    assert hid.shape == ref.shape and 0 < n < hid.shape[0]
    check(f"ragged_golden_{tag}/loss_rel", rel(out["loss"], g[f"{tag}_loss"]), 1e-3)
    check(f"ragged_golden_{tag}/hidden_real_rows_maxrel", max_rel(hid[:n], ref[:n]), 4e-2)
    check(f"ragged_golden_{tag}/hidden_padded_rows_maxrel", max_rel(hid[n:], ref[n:]), 4e-2)
    names = [("depth", 2), ("seg", 1), ("seg", 2), ("gen", 3)]
    for i, key in enumerate(names):
        mine = out["layer_losses"][key].float().cpu().numpy()
        check(f"ragged_golden_{tag}/layer_loss/{key[0]}@{key[1]}", _trip_err(mine, g[f"{tag}_layer_losses"][i]), 1e-2)


def test_edge_truncation():
    """ola_arch.py:394-397 truncates the spliced sequence (658 rows here) to tokenizer_model_max_length."""
    out, ref = _edge_case({"tokenizer_model_max_length": 652}, lambda b: None, tag="edge_trunc")      # keeps 8 supervised tokens per sample
    assert out["plan"]["S"] == 652 == ref["labels"].shape[1]


def test_edge_short_sequence_head_path():
    """S < 600 (text-only batch): forward_emb_predictor hands the WHOLE layer state to the heads (base_ola_vlm.py:419-421).
    (With "gen" in aux_mode the reference then slices an empty latent block and returns NaN, so this case uses depth-seg.)"""
    def mutate(b):
        b["input_ids"][:, 38] = 7
        b.pop("gen_target", None); b.pop("gen_mask", None)
    out, ref = _edge_case({"aux_mode": "depth-seg"}, mutate, tag="edge_short", min_cos=0.96, max_norm=5e-2)
    assert out["plan"]["S"] == 59


@pytest.mark.parametrize("arch", ["llama", "phi3"])
def test_ift_stage_llm_weight_gradients_match_oracle(arch):
    """SURVEY §8f f-2 (IFT / visual-instruction-tuning step: NTP only, the whole LLM + projector trainable, tower frozen;
    scripts/train/finetune.sh, llava_llama.py:73-119): every parameter gradient of the fused forward+backward against autograd
    through the fp32 oracle on the same bf16-rounded weights, then one AdamW step must lower the loss."""
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine, is_trainable
    ocfg, W, batch, g = cases.tiny_llama_case(arch)
    ocfg = O.make_config(**{**vars(ocfg), "aux_mode": "", "num_task_tokens": 0})
    batch = {k: v for k, v in batch.items() if not (k.endswith("_target") or k.endswith("_mask")) or k == "attention_mask"}
    eng = Engine(VisperConfig(**vars(ocfg), train_llm=True))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    tr = [k for k in eng.ps.index]
    assert all(is_trainable(k, True) for k in tr) and "lm_head.weight" in tr and "model.embed_tokens.weight" in tr
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    check(f"ift_{arch}/loss_rel", rel(out["loss"], ref["loss"]), 1e-3)
    for k in tr:
        mine = eng.ps.g(k).detach().float().cpu().reshape(-1)
        if Wq[k].grad is None:                                       # heads / task tokens of the fixture: unused without aux tasks
            assert float(mine.abs().max()) == 0.0, k
            continue
        c, n = grad_err(mine, Wq[k].grad.reshape(-1))
        check(f"ift_{arch}/grad/{k}/one_minus_cos", c, 1e-3)
        check(f"ift_{arch}/grad/{k}/norm_dev", n, 1.5e-2)
    l0 = float(out["loss"])
    for _ in range(3):
        eng.optimizer_step(lr=2e-4)
        l1 = float(eng.train_step(_to_gpu_batch(batch))["loss"])
    assert l1 < l0, (l0, l1)
    # the re-transposed dgrad copies follow the updated weights
    assert torch.equal(eng.fz["dec.0.wo_T"], eng.fz["dec.0.wo"].t().contiguous())


def test_ift_stage_matches_reference_golden():
    """Same step against the fixture produced by the reference's own LlavaLlamaForCausalLM (tests/golden/tiny_llama_ift.npz):
    loss within 1e-2 (bf16 vs the reference's fp32), every parameter-gradient norm within 10 %, subsampled gradients aligned."""
    from oracle import cases
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_ift_case()
    eng = Engine(VisperConfig(**vars(ocfg), train_llm=True))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    check("ift_golden/loss_rel_vs_reference_golden", rel(out["loss"], g["loss"]), 1e-3)
    tr = json.loads(str(g["trainable"]))
    assert sorted(eng.ps.index) == tr
    for k in tr:
        got = eng.ps.g(k).detach().float().cpu()
        ref_norm = float(g[f"gradnorm::{k}"])
        if ref_norm == 0.0:
            assert float(got.norm()) < 1e-7, k
        else:
            check(f"ift_golden/gradnorm/{k}_rel", abs(float(got.norm()) - ref_norm) / ref_norm, 2.5e-2)
        mine, theirs = torch.from_numpy(cases.sub(got, 128)), torch.from_numpy(g[f"gradsub::{k}"])
        if float(theirs.norm()) > 0:
            c, _ = grad_err(mine, theirs)
            check(f"ift_golden/gradsub/{k}/one_minus_cos", c, 2e-3)


def test_dinov2_depth_teacher_matches_oracle_and_reference_golden():
    """the batched DINOv2 depth teacher on the GPU (teachers.DinoV2DepthTeacher) against the fp32 oracle on the same
    bf16-rounded weights / images and against the reference's own DinoVisionTransformer output (tests/golden/dinov2_teacher.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases, visper_oracle as O, weights as WT
    from visper_lm_amd.teachers import DinoV2DepthTeacher
    g = cases.load_golden("dinov2_teacher.npz")
    man = json.loads(str(g["manifest"]))
    dims = json.loads(str(g["dims"]))
    taps = [int(t) for t in g["taps"]]
    W = cases.dinov2_weights(man)
    assert {k: tuple(v) for k, v in man.items()} == DinoV2DepthTeacher.shapes(dims["embed_dim"], dims["depth"])
    t = DinoV2DepthTeacher(dims["embed_dim"], dims["depth"], dims["num_heads"], taps)
    t.load_weights(W)
    images = WT.tensor("dino_images", (2, 3, 336, 336))
    got = t.forward(images.cuda()).float().cpu()
    with torch.no_grad():
        ref = O.dinov2_depth_target(images.to(BF).float(), {k: v.to(BF).float() for k, v in W.items()}, dims["num_heads"], taps)
    assert tuple(got.shape) == tuple(g["target_shape"])
    err = (got - ref).abs().max() / ref.abs().max()
    assert float(err) < 3e-2, float(err)
    gerr = np.abs(got[:, ::7, ::3].numpy() - g["target_sub"]).max() / np.abs(g["target_sub"]).max()
    assert float(gerr) < 4e-2, float(gerr)


def test_clip_image_embed_teacher_matches_oracle_and_hf_golden():
    """the batched generation teacher (teachers.ClipImageEmbedTeacher) against the fp32 oracle on bf16-rounded weights
    and against HF's CLIPVisionModelWithProjection output (tests/golden/clip_embed_teacher.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases, visper_oracle as O, weights as WT
    from visper_lm_amd.teachers import ClipImageEmbedTeacher
    g = cases.load_golden("clip_embed_teacher.npz")
    dims = json.loads(str(g["dims"]))
    W = {k: WT.param(k, s) for k, s in json.loads(str(g["manifest"])).items()}
    t = ClipImageEmbedTeacher(dims["hidden_size"], dims["num_hidden_layers"], dims["num_attention_heads"], dims["image_size"],
                              dims["patch_size"], act=dims["hidden_act"])
    t.load_weights(W)
    images = WT.tensor("clip_embed_images", (2, 3, 224, 224))
    got = t.forward(images.cuda()).float().cpu()
    with torch.no_grad():
        ref = O.clip_image_embeds(images.to(BF).float(), {k: v.to(BF).float() for k, v in W.items()}, dims["num_attention_heads"],
                                  dims["patch_size"], act=dims["hidden_act"])
    assert tuple(got.shape) == (2, 1, dims["projection_dim"])
    assert float((got - ref).abs().max() / ref.abs().max()) < 3e-2
    assert float(np.abs(got.numpy() - g["embeds"]).max() / np.abs(g["embeds"]).max()) < 4e-2


def test_swin_seg_teacher_matches_oracle_and_hf_golden():
    """the batched segmentation teacher (teachers.SwinSegTeacher: window attention with relative-position bias and
    shift masks through vp_attn_fwd_bias) against the fp32 oracle on bf16-rounded weights and HF's SwinBackbone golden."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases, visper_oracle as O, weights as WT
    from visper_lm_amd.teachers import SwinSegTeacher
    g = cases.load_golden("swin_teacher.npz")
    dims = json.loads(str(g["dims"]))
    W = cases.swin_weights(json.loads(str(g["manifest"])))
    t = SwinSegTeacher(dims["embed_dim"], dims["depths"], dims["num_heads"], dims["window_size"], dims["patch_size"], dims["image_size"])
    t.load_weights(W)
    images = WT.tensor("swin_images", (2, 3, 384, 384))
    got = t.forward(images.cuda()).float().cpu()
    with torch.no_grad():
        ref = O.swin_seg_target(images.to(BF).float(), {k: v.to(BF).float() for k, v in W.items()}, dims["depths"], dims["num_heads"],
                                window=dims["window_size"], patch=dims["patch_size"])
    assert tuple(got.shape) == tuple(g["target_shape"])
    assert float((got - ref).abs().max() / ref.abs().max()) < 4e-2
    assert float(np.abs(got[:, ::5, ::3, ::3].numpy() - g["target_sub"]).max() / np.abs(g["target_sub"]).max()) < 5e-2


def test_clip_image_embed_teacher_padded_head_dim():
    """ViT-H has head_dim 80, which is not a kernel head size: the teacher zero-pads every head to 96 inside the frozen weights.
    Checked against the fp32 oracle at hidden 160 / 2 heads (head_dim 80)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import visper_oracle as O, weights as WT
    from visper_lm_amd.teachers import ClipImageEmbedTeacher
    C, L, nh, S, P, proj = 160, 2, 2, 112, 14, 64
    pre = "pipe.image_encoder."
    sh = {pre + "vision_model.embeddings.class_embedding": (C,), pre + "vision_model.embeddings.patch_embedding.weight": (C, 3, P, P),
          pre + "vision_model.embeddings.position_embedding.weight": ((S // P) ** 2 + 1, C), pre + "visual_projection.weight": (proj, C)}
    for n in ("pre_layrnorm", "post_layernorm"):
        sh[pre + f"vision_model.{n}.weight"] = (C,); sh[pre + f"vision_model.{n}.bias"] = (C,)
    for l in range(L):
        q = pre + f"vision_model.encoder.layers.{l}."
        for x in ("q", "k", "v", "out"):
            sh[q + f"self_attn.{x}_proj.weight"] = (C, C); sh[q + f"self_attn.{x}_proj.bias"] = (C,)
        for n in ("layer_norm1", "layer_norm2"):
            sh[q + n + ".weight"] = (C,); sh[q + n + ".bias"] = (C,)
        sh[q + "mlp.fc1.weight"] = (4 * C, C); sh[q + "mlp.fc1.bias"] = (4 * C,)
        sh[q + "mlp.fc2.weight"] = (C, 4 * C); sh[q + "mlp.fc2.bias"] = (C,)
    W = {k: WT.param(k, s) for k, s in sh.items()}
    t = ClipImageEmbedTeacher(C, L, nh, S, P)
    t.load_weights(W)
    assert t.hp == 96
    images = WT.tensor("clip_pad_images", (2, 3, S, S))
    got = t.forward(images.cuda()).float().cpu()
    with torch.no_grad():
        ref = O.clip_image_embeds(images.to(BF).float(), {k: v.to(BF).float() for k, v in W.items()}, nh, P)
    assert float((got - ref).abs().max() / ref.abs().max()) < 3e-2


def test_left_padding_ntp_matches_oracle():
    """ola_arch.py:408-427 with tokenizer_padding_side == "left" (ragged batch, NTP only: aux heads + ragged left padding are refused,
    see splice.host_plan): loss and every gradient equal the oracle's left-padded run; labels / attention_mask / position_ids and the
    real rows of inputs_embeds / hidden are presented right-aligned exactly like the reference lays them out."""
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    from parity import check, rel, max_rel, grad_err
    ocfg, W, batch, g = cases.tiny_llama_case()
    ocfg = O.make_config(**{**vars(ocfg), "aux_mode": "", "num_task_tokens": 0, "tokenizer_padding_side": "left"})
    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items() if k in ("input_ids", "labels", "attention_mask", "images")}
    batch["attention_mask"][1, 42:] = False
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    tr = [k for k in eng.ps.index]
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    plan = out["plan"]
    assert plan["side"] == "left" and not plan["full"]
    assert torch.equal(plan["labels"], ref["labels"])
    am = plan["attention_mask"]
    check("left_pad/loss_rel", rel(out["loss"], ref["loss"]), 1e-3)
    emb, remb = out["inputs_embeds"].float().cpu(), ref["inputs_embeds"].detach()
    assert float(emb[~am].abs().max()) == 0.0 and float(remb[~am].abs().max()) == 0.0          # pad rows are zeros on the left
    check("left_pad/inputs_embeds_maxrel", max_rel(emb[am], remb[am]), 1.5e-2)
    check("left_pad/hidden_real_rows_maxrel", max_rel(out["hidden"].float().cpu()[am], ref["hidden"].detach()[am]), 3e-2)
    for k in tr:
        want = Wq[k].grad
        got = eng.ps.g(k).detach().float().cpu()
        if want is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        c, n = grad_err(got, want)
        check(f"left_pad/grad/{k}/one_minus_cos", c, 1e-3)
        check(f"left_pad/grad/{k}/norm_dev", n, 1e-2)


def test_checkpoint_resume_is_bitwise(tmp_path):
    """the PT run must persist heads / task tokens / logit scales and the optimizer state (llava_trainer.py:997-1016 +
    HF Trainer checkpoints, resume at ola_vlm_train.py:1306-1309): 2 steps + save + load into a FRESH engine + 2 steps ==
    4 uninterrupted steps, bit for bit; weights-only loading gives the bf16-rounded parameters."""
    from oracle import cases
    from safetensors.torch import load_file
    from visper_lm_amd import data
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_llama_case()
    b = _to_gpu_batch(batch)

    def fresh():
        e = Engine(VisperConfig(**vars(ocfg)))
        e.load_weights(W)
        return e

    def steps(e, n):
        for _ in range(n):
            e.train_step(b)
            e.optimizer_step(lr=1e-3, weight_decay=0.01)
    a = fresh(); steps(a, 4)
    c = fresh(); steps(c, 2)
    data.save_checkpoint(c, str(tmp_path))
    sd = load_file(str(tmp_path / "trainable.safetensors"))
    assert sorted(sd) == sorted(c.ps.index) and any("_heads." in k for k in sd) and "seg_logit_scale" in sd
    assert sd["model.special_depth_tokens"].dtype == torch.bfloat16 and sd["seg_logit_scale"].dtype == torch.float32
    d = fresh()
    loaded = data.load_checkpoint(d, str(tmp_path))
    assert sorted(loaded) == sorted(d.ps.index) and d.ps.step == 2
    steps(d, 2)
    assert torch.equal(a.ps.master, d.ps.master) and torch.equal(a.ps.exp_avg_sq, d.ps.exp_avg_sq) and torch.equal(a.ps.shadow, d.ps.shadow)
    e = fresh()
    data.load_checkpoint(e, str(tmp_path), resume_optimizer=False)
    assert e.ps.step == 0 and torch.equal(e.ps.shadow, c.ps.shadow)


def test_resampler_depth_two_heads_match_oracle():
    """resampler.py:217-219 loops over `depth` Perceiver blocks (the scripts use 1): depth = 2 for all three heads, HIP vs the fp32 oracle
    (itself pinned against the reference's TaskTokenResampler(depth=2): tests/golden/units.npz rs_deep)."""
    from oracle import cases, visper_oracle as O, weights as WT
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    from visper_lm_amd.params import param_shapes
    ocfg, W, batch, g = cases.tiny_llama_case()
    kw = dict(vars(ocfg))
    for k in ("image_gen", "image_seg", "image_depth"):
        kw[k] = dict(kw[k], depth=2)
    ocfg2 = O.make_config(**kw)
    cfg = VisperConfig(**kw)
    W = dict(W)
    for k, s in param_shapes(cfg, vit_nested=False).items():
        if k not in W:
            W[k] = WT.param(k, s)                                     # the second block's parameters
    eng = Engine(cfg)
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    tr = [k for k in eng.ps.index]
    assert any(".layers.1.0.to_q.weight" in k for k in tr)
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg2)
    ref["loss"].backward()
    check("depth2/loss_rel", rel(out["loss"], ref["loss"]), 1e-3)
    for key, trip in ref["layer_losses"].items():
        check(f"depth2/layer_loss/{key[0]}@{key[1]}", _trip_err(out["layer_losses"][key].float().cpu().numpy(), [float(x) for x in trip]), 1e-2)
    for k in tr:
        want = Wq[k].grad
        got = eng.ps.g(k).detach().float().cpu()
        if want is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        if got.numel() == 1:
            continue
        c, n = grad_err(got, want)
        check(f"depth2/grad/{k}/one_minus_cos", c, 3e-2)
        check(f"depth2/grad/{k}/norm_dev", n, 6e-2)


# ------------------------------------------------------------------------------------------------ round 3: VERDICT r2 missing-2 / -5 / -6 / -7
def test_ift_stage_with_pt_task_tokens_matches_reference_golden():
    """The reference's own PT -> IFT hand-off (scripts/train/finetune.sh on a PT checkpoint): LlavaLlamaForCausalLM with num_task_tokens 8 and
    task_token_format "emb" splices the RAW (576, H) depth / seg parameters + 8 gen rows behind the image (llava_arch.py:250-293).  Engine
    (task_token_layout "raw") against tests/golden/tiny_llama_ift_tok.npz from the reference itself, and against the fp32 oracle."""
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_ift_tok_case()
    eng = Engine(VisperConfig(**vars(ocfg), train_llm=True))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    assert out["plan"]["S"] == 58 + 576 * 3 + 8 == int(g["logits_shape"][1])
    check("ift_tok/loss_rel_vs_reference_golden", rel(out["loss"], g["loss"]), 1e-3)
    tr = json.loads(str(g["trainable"]))
    assert sorted(eng.ps.index) == tr and len(tr) == 46
    for k in tr:
        got = eng.ps.g(k).detach().float().cpu()
        ref_norm = float(g[f"gradnorm::{k}"])
        check(f"ift_tok/gradnorm/{k}_rel", abs(float(got.norm()) - ref_norm) / ref_norm, 2.5e-2)
        mine, theirs = torch.from_numpy(cases.sub(got, 128)), torch.from_numpy(g[f"gradsub::{k}"])
        if float(theirs.norm()) > 0:
            c, _ = grad_err(mine, theirs)
            check(f"ift_tok/gradsub/{k}/one_minus_cos", c, 2e-3 if "special_" not in k else 5e-3)
    # full token-parameter gradients (not just a subsample) against the fp32 oracle on the same bf16-rounded weights
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    for t in ("depth", "seg", "gen"):
        k = f"model.special_{t}_tokens"
        c, n = grad_err(eng.ps.g(k).detach().float().cpu().reshape(-1), Wq[k].grad.reshape(-1))
        check(f"ift_tok/oracle/{k}/one_minus_cos", c, 2e-3)
        check(f"ift_tok/oracle/{k}/norm_dev", n, 2e-2)
    # the pooled layout ("expand_emb", llava_arch.py:261-263) on the same weights gives the PT-stage sequence length
    import copy
    oc2 = copy.copy(ocfg); oc2.task_token_layout = "pooled"
    eng2 = Engine(VisperConfig(**vars(oc2), train_llm=True))
    eng2.load_weights(W)
    out2 = eng2.train_step(_to_gpu_batch(batch))
    ref2 = O.forward({k: v.to(BF).float() for k, v in W.items()}, bq, oc2)
    assert out2["plan"]["S"] == 58 + 576 + 24
    check("ift_tok/expand_emb/loss_rel", rel(out2["loss"], ref2["loss"]), 1e-3)


def test_pt_step_without_task_tokens_matches_oracle_and_reference_golden():
    """num_task_tokens == 0: GenHead / DepthHead / OneFormerSegHead around the plain Resampler with its own `latents` parameter, whole layer
    state as head input (base_ola_vlm.py:120-169, 420-422, 429-430; resampler.py:120-165).  Engine vs tests/golden/tiny_llama_nt0.npz
    (the reference itself, fp32) and vs the fp32 oracle on the same bf16-rounded weights (every trainable gradient)."""
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_nt0_case()
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    assert out["plan"]["S"] == 58 + 576 and out["plan"]["n_tok_rows"] == 0
    check("nt0/loss_rel_vs_reference_golden", rel(out["loss"], g["loss"]), 1e-3)
    order = [("depth", 2), ("seg", 1), ("seg", 2), ("gen", 3)]
    for i, key in enumerate(order):
        check(f"nt0/layer_loss_vs_golden/{key[0]}@{key[1]}", _trip_err(out["layer_losses"][key].float().cpu().numpy(), g["layer_losses"][i]), 1.2e-2)
    tr = json.loads(str(g["trainable"]))
    assert sorted(eng.ps.index) == tr and sum(k.endswith("projector.latents") for k in tr) == 4
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    check("nt0/loss_rel_vs_oracle", rel(out["loss"], ref["loss"]), 1e-3)
    for k in tr:
        got, want = eng.ps.g(k).detach().float().cpu().reshape(-1), Wq[k].grad
        if want is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        if got.numel() == 1:
            check(f"nt0/grad/{k}_abs", abs(float(got) - float(want)), 0.05 * abs(float(want)) + 1e-3)
            continue
        c, n = grad_err(got, want.reshape(-1))
        check(f"nt0/grad/{k}/one_minus_cos", c, 1.5e-2)
        check(f"nt0/grad/{k}/norm_dev", n, 3e-2)


def test_pt_step_without_intermediate_depth_matches_oracle_and_reference_golden():
    """image_depth["use_intermediate_depth"] = False (VERDICT r5 missing-2; base_ola_vlm.py:132,462-466, da_v2_head.py:437-455): the depth head has
    no linear_1..3 parameters, its loss compares visual_feats itself, depth_embs entries hold one map and the DPT decoder runs on [feats[0]] * 4.
    Engine vs tests/golden/tiny_llama_noid.npz (the reference itself, fp32) and vs the fp32 oracle on the same bf16-rounded weights."""
    from oracle import cases, visper_oracle as O
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_noid_case()
    eng = Engine(VisperConfig(**vars(ocfg), depth_decoder=True))
    eng.load_weights(W)
    out = eng.train_step(_to_gpu_batch(batch))
    tr = json.loads(str(g["trainable"]))
    assert sorted(eng.ps.index) == tr and not any(".linear_" in k for k in eng.ps.index)
    check("noid/loss_rel_vs_reference_golden", rel(out["loss"], g["loss"]), 1e-3)
    order = [("depth", 2), ("seg", 1), ("seg", 2), ("gen", 3)]
    for i, key in enumerate(order):
        check(f"noid/layer_loss_vs_golden/{key[0]}@{key[1]}", _trip_err(out["layer_losses"][key].float().cpu().numpy(), g["layer_losses"][i]), 1.2e-2)
    assert len(out["depth_feats"][0]) == 1 == int(g["depth_embs_len"]) and tuple(out["depth_feats"][0][0].shape) == (2, 576, 1024)
    dp = out["depth_preds"][0].float().cpu()
    assert tuple(dp.shape) == tuple(int(v) for v in g["depth_preds_shape"])
    check("noid/depth_pred_mean_abs_vs_reference_golden", np.abs(dp[:, ::5, ::5].numpy() - g["depth_pred_sub"]).mean(), 2e-2)
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    for k in tr:
        Wq[k] = Wq[k].clone().requires_grad_(True)
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    check("noid/loss_rel_vs_oracle", rel(out["loss"], ref["loss"]), 1e-3)
    for k in tr:
        got, want = eng.ps.g(k).detach().float().cpu().reshape(-1), Wq[k].grad
        if want is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        if got.numel() == 1:
            check(f"noid/grad/{k}_abs", abs(float(got) - float(want)), 0.05 * abs(float(want)) + 1e-3)
            continue
        c, n = grad_err(got, want.reshape(-1))
        check(f"noid/grad/{k}/one_minus_cos", c, 1.5e-2)
        check(f"noid/grad/{k}/norm_dev", n, 3e-2)


def test_list_and_5d_images_flat_merge(tiny):
    """VERDICT r5 missing-3: `images` as a list or a 5-D tensor (ola_arch.py:262-275, mm_patch_merge_type "flat").  A list of [3, H, W] tensors is
    the stacked 4-D batch (bit-identical step).  A 5-D [B, 2, 3, H, W] batch: both images of a sample are encoded, their features flattened to
    2 x 576 rows behind ONE <image> token, then the task tokens: loss, layer losses and every gradient against the fp32 oracle's own flat merge.
    Anything but the flat merge is refused with the reference lines."""
    from oracle import visper_oracle as O
    eng, batch, ocfg, tr = tiny["eng"], tiny["batch"], tiny["ocfg"], tiny["tr"]
    gb = _to_gpu_batch(batch)
    lst = dict(gb, images=[im for im in gb["images"]])
    out = eng.train_step(lst)
    assert torch.equal(out["loss"], tiny["out"]["loss"]) and torch.equal(out["logits"], tiny["out"]["logits"])
    for k in eng.ps.index:
        assert torch.equal(eng.ps.g(k).detach().float().cpu(), tiny["grads"][k]), k
    im5 = torch.stack([batch["images"], batch["images"].flip(0) * 0.5], 1)                     # [B, 2, 3, H, W]
    out5 = eng.train_step(dict(gb, images=im5.cuda()))
    assert out5["plan"]["S"] == tiny["out"]["plan"]["S"] + 576 and out5["plan"]["n_feat"] == 4 * 576
    Wq = {k: (v.detach().clone().requires_grad_(True) if k in tr else v.detach()) for k, v in tiny["Wq"].items()}
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    bq["images"] = im5.to(BF).float()
    ref = O.forward(Wq, bq, ocfg)
    ref["loss"].backward()
    check("flat5d/loss_rel_vs_oracle", rel(out5["loss"], ref["loss"]), 1e-3)
    check("flat5d/text_loss_rel_vs_oracle", rel(out5["text_loss"], ref["text_loss"]), 1e-3)
    check("flat5d/inputs_embeds_maxrel", max_rel(out5["inputs_embeds"].cpu(), ref["inputs_embeds"].detach()), 1.5e-2)
    for key, trip in ref["layer_losses"].items():
        check(f"flat5d/layer_loss/{key[0]}@{key[1]}_vs_oracle", _trip_err(out5["layer_losses"][key].float().cpu().numpy(), np.array([float(x) for x in trip])), 1.2e-2)
    for k in tr:
        got, want = eng.ps.g(k).detach().float().cpu().reshape(-1), Wq[k].grad
        if want is None or got.numel() == 1:
            continue
        c, n = grad_err(got, want.reshape(-1))
        check(f"flat5d/grad/{k}/one_minus_cos", c, 1.5e-2)
    eng.cfg.mm_patch_merge_type = "spatial_unpad"
    try:
        with pytest.raises(NotImplementedError, match="ola_arch.py"):
            eng.train_step(lst)
    finally:
        del eng.cfg.mm_patch_merge_type
    with pytest.raises(IndexError):
        eng.train_step(dict(gb, images=[gb["images"][0]]))                                     # fewer entries than <image> tokens


def test_emb_loss_batch_repeat_branch_in_the_step():
    """_emb_loss's repeat branch (base_ola_vlm.py:292-299): ONE gen / depth target row for the batch of two predictions -> targets and
    masks tiled.  The rank-4 seg target cannot take that branch (the reference's 3-argument repeat raises): ValueError here."""
    def mutate(b):
        b["gen_target"], b["gen_mask"] = b["gen_target"][:1].clone(), torch.tensor([0.5])
        b["depth_target"], b["depth_mask"] = b["depth_target"][:1].clone(), torch.tensor([1.0])
    out, ref = _edge_case({}, mutate, tag="edge_repeat")
    from oracle import cases
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_llama_case()
    batch = dict(batch, seg_target=batch["seg_target"][:1].clone())
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    with pytest.raises(ValueError):
        eng.train_step(_to_gpu_batch(batch))
