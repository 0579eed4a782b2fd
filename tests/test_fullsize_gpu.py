"""Parity at BASELINE.json's full sizes (configs[1]: B=8, S=2048, Llama-3-8B dims) through size-independent properties: the oracle
cannot run these shapes in seconds, so kernels are checked on row samples against fp32 math, by linearity / determinism, and the
fused kernels against their unfused compositions."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visper_lm_amd import ops as o
    return o


def _rnd(*shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(BF)


@pytest.mark.parametrize("M,N,K,res", [(16384, 4096, 4096, True), (16384, 6144, 4096, False), (16384, 4096, 14336, True),
                                       (8192, 128256, 4096, False)])
def test_gemm_full_size_row_samples(ops, M, N, K, res):
    """The persistent 8-phase kernel at the train step's shapes: 384 sampled rows against fp32 math with the reference's rounding
    points (linear -> bf16, + residual -> bf16), every tile column visited; plus bitwise run-to-run determinism."""
    a, w = _rnd(M, K, seed=1), _rnd(N, K, seed=2, scale=0.02)
    r = _rnd(M, N, seed=3) if res else None
    out = ops.gemm(a, w, residual=r)
    rows = torch.cat([torch.arange(0, 128), torch.arange(M // 2 - 64, M // 2 + 64), torch.arange(M - 128, M)]).cuda()
    ref = (a[rows].float() @ w.float().t()).to(BF).float()
    if res:
        ref = (ref + r[rows].float()).to(BF).float()
    err = (out[rows].float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 1e-2, float(err)
    assert torch.equal(out, ops.gemm(a, w, residual=r))


def test_fused_swiglu_full_size_equals_unfused(ops):
    M, H, Fd = 16384, 4096, 14336
    x, wgu, wdT, dy = _rnd(M, H, seed=4), _rnd(2 * Fd, H, seed=5, scale=0.02), _rnd(Fd, H, seed=6, scale=0.02), _rnd(M, H, seed=7)
    gu, act = ops.gemm_swiglu_fwd(x, wgu)
    gu_ref = ops.gemm(x, wgu)
    assert torch.equal(gu, gu_ref) and torch.equal(act, ops.swiglu_fwd(gu_ref))
    dgu, dgu_ref = ops.gemm_swiglu_bwd(dy, wdT, gu), ops.swiglu_bwd(ops.gemm(dy, wdT), gu)
    diff = (dgu.float() - dgu_ref.float()).abs()
    assert float((diff > 0).float().mean()) < 1e-4            # isolated 1-ulp FMA-contraction flips only
    assert float((diff / dgu_ref.float().abs().clamp_min(1e-3)).max()) < 1e-2


def test_attention_full_size_rows_linearity_determinism(ops):
    """B=8, 32 q / 8 kv heads, S=2048, D=128 causal: sampled (batch, head) slices against fp32 softmax attention; backward is
    linear in dO (bwd(a*dO1 + b*dO2) = a*bwd(dO1) + b*bwd(dO2) up to bf16 rounding) and both directions are deterministic."""
    B, Hq, Hkv, S, D = 8, 32, 8, 2048, 128
    qkv = _rnd(B, S, (Hq + 2 * Hkv) * D, seed=8)
    q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
    o, lse = ops.attn_fwd(q, k, v, True)
    o2, _ = ops.attn_fwd(q, k, v, True)
    assert torch.equal(o, o2)
    mask = torch.ones(S, S, device="cuda", dtype=torch.bool).tril()
    for b, h in ((0, 0), (3, 13), (7, 31)):
        qs, ks, vs = q[b, :, h].float(), k[b, :, h // 4].float(), v[b, :, h // 4].float()
        p = torch.softmax((qs @ ks.t() / D ** 0.5).masked_fill(~mask, float("-inf")), -1)
        ref = p @ vs
        assert float((o[b, :, h].float() - ref).abs().max() / ref.abs().max()) < 2e-2
    d1, d2 = _rnd(B, S, Hq, D, seed=9), _rnd(B, S, Hq, D, seed=10)
    g1 = ops.attn_bwd(q, k, v, o, lse, d1, True)
    g2 = ops.attn_bwd(q, k, v, o, lse, d2, True)
    g12 = ops.attn_bwd(q, k, v, o, lse, (0.5 * d1.float() - 2.0 * d2.float()).to(BF), True)
    for a_, b_, c_ in zip(g1, g2, g12):
        want = 0.5 * a_.float() - 2.0 * b_.float()
        assert float((c_.float() - want).abs().max() / want.abs().max()) < 3e-2
    g1b = ops.attn_bwd(q, k, v, o, lse, d1, True)
    assert all(torch.equal(x, y) for x, y in zip(g1, g1b))
    # fp32 reference gradients on one (batch, kv head) group: dK / dV sum over the 4 q heads of the group
    b, hk = 2, 5
    qs = q[b, :, hk * 4:hk * 4 + 4].float().permute(1, 0, 2).requires_grad_(True)          # [4, S, D]
    ks, vs = k[b, :, hk].float().requires_grad_(True), v[b, :, hk].float().requires_grad_(True)
    p = torch.softmax((qs @ ks.t() / D ** 0.5).masked_fill(~mask, float("-inf")), -1)
    (p @ vs).backward(d1[b, :, hk * 4:hk * 4 + 4].float().permute(1, 0, 2))
    dq, dk, dv = g1
    for got, want in ((dq[b, :, hk * 4:hk * 4 + 4].float().permute(1, 0, 2), qs.grad), (dk[b, :, hk].float(), ks.grad), (dv[b, :, hk].float(), vs.grad)):
        assert float((got - want).abs().max() / want.abs().max()) < 3e-2


def test_train_step_full_size_is_deterministic_and_finite():
    """One configs[1] step twice from the same state: bitwise identical loss and gradients (no atomics anywhere on the path)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from visper_lm_amd.config import llama3_8b
    from visper_lm_amd.engine import Engine
    cfg = llama3_8b(num_hidden_layers=4)
    cfg.image_gen["img_layer_indices"] = "4"; cfg.image_depth["depth_layer_indices"] = "3"; cfg.image_seg["seg_layer_indices"] = "3"
    eng = Engine(cfg)
    eng.init_random(0)
    batch = bench.make_batch(cfg, 8, 1449, 0, torch.device("cuda"))
    o1 = eng.train_step(batch); g1 = eng.ps.grad.clone(); l1 = o1["loss"].clone()
    o2 = eng.train_step(batch)
    assert o1["plan"]["S"] == 2048 and torch.isfinite(l1).all() and torch.isfinite(g1).all()
    assert torch.equal(l1, o2["loss"]) and torch.equal(g1, eng.ps.grad)


def test_convnext_xxl_tower_full_size_finite_and_deterministic():
    """BASELINE configs[3] at its real size: CLIP-ConvNeXt-XXL trunk (dims 384/768/1536/3072, depths 3/4/30/3, 768 px -> 576 x 3072;
    clip_convnext_encoder.py:92-101,150-174) on the HIP path — shapes, finiteness, bitwise run-to-run determinism, and the spatial
    structure the trunk must keep (rows are (y, x) raster order: a constant image gives identical interior tokens).  The tower stays
    "parity unpinned" against timm (absent); tests/test_convnext_pin.py pins the restatement piecewise against stock torch.nn modules."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visper_lm_amd.config import llama3_8b_convnext
    from visper_lm_amd.engine import Engine
    cfg = llama3_8b_convnext(num_hidden_layers=1, aux_mode="", num_task_tokens=0, vocab_size=1024)
    eng = Engine(cfg)
    eng.init_random(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    images = torch.randn(2, 3, 768, 768, device="cuda", generator=g).to(BF)
    f1 = eng.vit_forward(images)
    f2 = eng.vit_forward(images)
    assert tuple(f1.shape) == (2 * 576, 3072) and torch.isfinite(f1.float()).all() and torch.equal(f1, f2)
    assert float(f1.float().std()) > 0
    const = torch.full((1, 3, 768, 768), 0.5, device="cuda", dtype=BF)
    fc = eng.vit_forward(const).view(24, 24, 3072).float()
    inner = fc[6:18, 6:18].reshape(-1, 3072)                 # receptive fields entirely inside the image see the same input
    assert float((inner - inner[0]).abs().max()) <= 1e-2 * float(inner.abs().max())


def _two_steps_bitwise(cfg, B, T, S_expect):
    import bench
    from visper_lm_amd.engine import Engine
    eng = Engine(cfg)
    eng.init_random(0)
    batch = bench.make_batch(cfg, B, T, 0, torch.device("cuda"))
    o1 = eng.train_step(batch); g1 = eng.ps.grad.clone(); l1 = o1["loss"].clone()
    ll1 = {k: v.clone() for k, v in o1["layer_losses"].items()}
    o2 = eng.train_step(batch)
    torch.cuda.synchronize()
    assert o1["plan"]["S"] == S_expect, o1["plan"]["S"]
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all() and float(g1.norm()) > 0
    assert len(ll1) == 3 and all(torch.isfinite(v).all() for v in ll1.values())
    assert torch.equal(l1, o2["loss"]) and torch.equal(g1, eng.ps.grad)
    assert all(torch.equal(ll1[k], o2["layer_losses"][k]) for k in ll1)
    return o1


def test_config4_phi3_full_depth_step_is_deterministic_and_finite():
    """BASELINE configs[4] ITSELF, the step `bench.py --workload phi3` times: CLIP-ViT-L + Phi-3-mini-4k at full depth (32 layers, 32 MHA heads x
    96, fused qkv / gate_up, sliding window 2047 inclusive), B=4, T=3497 -> post-splice S=4096 > window, three distillation heads: S as configured,
    finite losses / per-layer losses / gradients, two steps from the same state bitwise identical (ola_phi3.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visper_lm_amd.config import phi3_mini
    cfg = phi3_mini()
    out = _two_steps_bitwise(cfg, 4, 3497, 4096)
    assert out["plan"]["n_valid"] == 4 * (3497 - 1 - (cfg.num_sys_tokens + 6))


def test_config3_convnext_full_step_is_deterministic_and_finite():
    """BASELINE configs[3] ITSELF, the step `bench.py --workload convnext` times: CLIP-ConvNeXt-XXL at 768 px (576 x 3072 features,
    clip_convnext_encoder.py:92-101,150-174) -> projector 3072 -> 4096 -> Llama-3-8B at full depth, B=8, S=2048, three distillation heads."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visper_lm_amd.config import llama3_8b_convnext
    cfg = llama3_8b_convnext()
    out = _two_steps_bitwise(cfg, 8, 1449, 2048)
    assert tuple(out["image_features"].shape)[-1] == 4096


@pytest.mark.parametrize("stage", ["ift", "pt"])
def test_overlapped_steps_equal_synchronised_steps_full_width(stage):
    """Side-stream schedule at full width (2 decoder layers): with the host running ahead, step t+1's frozen tower (and step t's heads) execute beside
    the main stream's kernels — in the IFT stage beside the decoder FORWARD and its low-register RMSNorm / element-wise kernels.  Regression test of the
    round-4 fault (a 4-wave GEMM wave that did not own its SIMD's register file produced NaNs next to a foreign wave): six back-to-back steps without
    any synchronisation give bitwise the losses of the same six steps with a device synchronise after each."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from visper_lm_amd.config import llama3_8b
    from visper_lm_amd.engine import Engine
    T = 1473 if stage == "ift" else 1449

    def run(sync):
        cfg = llama3_8b(aux_mode="", num_task_tokens=0, train_llm=True) if stage == "ift" else llama3_8b()
        cfg.num_hidden_layers = 2
        cfg.depth_decoder = stage == "pt"
        if stage == "pt":
            cfg.image_gen["img_layer_indices"] = "2"; cfg.image_depth["depth_layer_indices"] = "2"; cfg.image_seg["seg_layer_indices"] = "2"
        eng = Engine(cfg, device=torch.device("cuda:0"))
        eng.set_distributed(0, 1, transport="torch")
        eng.init_random(seed=0)
        pool = [bench.make_batch(cfg, 8, T, 1000 * j, torch.device("cuda:0")) for j in range(2)]
        torch.cuda.synchronize()
        gi = torch.Generator().manual_seed(99)
        losses = []
        for it in range(6):
            b = dict(pool[it % 2])
            ids = torch.randint(0, 1000, (8, T), generator=gi)
            ids[:, cfg.num_sys_tokens] = -200
            lab = ids.clone()
            lab[:, :cfg.num_sys_tokens + 7] = -100
            b["input_ids"], b["labels"], b["images_resident"] = ids, lab, True
            out = eng.train_step(b)
            losses.append(out["loss"].clone())
            eng.optimizer_step(lr=1e-3, lr_mult=1.0)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res = torch.cat([l.reshape(1) for l in losses]).cpu()
        del eng, pool
        torch.cuda.empty_cache()
        return res

    a, b_ = run(True), run(False)
    assert torch.isfinite(a).all() and torch.isfinite(b_).all(), (a, b_)
    assert torch.equal(a, b_), (a, b_)


def test_attention_d128_generic_forward_matches_default():
    """The generic 16-row forward (VP_ATTN_FWDM=0) against the default 32x32x16 swapped-product kernel on the decoder shape: same lse and
    outputs up to bf16 rounding (different max bookkeeping), incl. ragged kv_len and a sliding window."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["VP_ROOT"])
from visper_lm_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B, Hq, Hkv, S, D = 2, 8, 2, 1000, 128
q = torch.randn(B, S, Hq, D, device="cuda", generator=g).bfloat16(); k = torch.randn(B, S, Hkv, D, device="cuda", generator=g).bfloat16()
v = torch.randn(B, S, Hkv, D, device="cuda", generator=g).bfloat16()
kv = torch.tensor([1000, 777], device="cuda", dtype=torch.int32)
outs = []
for w in (0, 300):
    o, lse = ops.attn_fwd(q, k, v, True, window=w, kv_len=kv)
    outs.append((o.float().cpu(), lse.cpu()))
torch.save(outs, os.environ["VP_OUT"])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("0", "1"):
        out = f"/tmp/vp_attn_ab_{flag}.pt"
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VP_ROOT=root, VP_OUT=out, VP_ATTN_FWDM=flag),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = torch.load(out)
    for (o0, l0), (o1, l1) in zip(res["0"], res["1"]):
        valid = torch.isfinite(l0) & (l0 > -1e29)
        assert float((o0 - o1).abs().max()) <= 2e-2 * float(o0.abs().max())
        assert float((l0[valid] - l1[valid]).abs().max()) < 1e-3


@pytest.mark.slow
def test_config1_real_step_vs_reference_style_bf16_cpu_path():
    """BASELINE.json configs[1] ITSELF — the step bench.py times: CLIP-ViT-L + Llama-3-8B, all 32 layers, B=8, T=1449 -> S=2048, three
    distillation heads (depth@18, seg@18, gen@20) — on identical random-init weights and batch against the oracle executed like the
    reference's CPU PyTorch path (bf16 weights and activations, PyTorch bf16 ops, eager attention, every row through lm_head; forward
    only, torch.no_grad).  Asserted (north-star: "NTP logits and per-layer embedding losses match the reference CPU PyTorch path within
    1e-3 bf16 relative"): total loss and NTP loss 1e-3; each layer's (emb, sl1, contrastive) triple — sl1 1e-3, emb / contrastive 1e-2
    (the contrastive term is a softmax over 8 cosine logits x exp(2), the noisiest scalar of the step; configs[0] measured 1.9e-3).
    ~3-6 min of host time on 64 threads: marked slow, inside the driver's GPU-test budget."""
    import os
    import time
    from oracle import visper_oracle as O
    from parity import check, rel
    from visper_lm_amd.config import llama3_8b
    from visper_lm_amd.engine import Engine
    from visper_lm_amd.params import param_shapes, init_value
    cfg = llama3_8b()
    B, T = 8, 1449
    gen = torch.Generator(device="cuda").manual_seed(0)
    W = {k: init_value(k, s, gen, torch.device("cuda"), BF if len(s) else torch.float32) for k, s in param_shapes(cfg, vit_nested=True).items()
         if not k.startswith("da_v2_head.")}
    g = torch.Generator().manual_seed(11)
    ns = cfg.num_sys_tokens
    ids = torch.randint(0, 1000, (B, T), generator=g)
    ids[:, ns] = -200
    labels = ids.clone()
    labels[:, :ns + 7] = -100
    rn = lambda *s: torch.randn(*s, generator=g).to(BF)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids, dtype=torch.bool), images=rn(B, 3, 336, 336),
                 gen_target=rn(B, 1, 1024), gen_mask=torch.ones(B), depth_target=rn(B, 576, 1024), depth_mask=torch.ones(B),
                 seg_target=rn(B, 1536, 24, 24), seg_mask=torch.ones(B))
    eng = Engine(cfg)
    eng.load_weights(W)
    out = eng.train_step({k: (v.cuda() if (k == "images" or k.endswith("_target") or k.endswith("_mask")) else v) for k, v in batch.items()})
    torch.cuda.synchronize()
    assert out["plan"]["S"] == 2048 and out["plan"]["n_valid"] == B * (T - 1 - (ns + 6))
    mine = dict(loss=float(out["loss"]), text=float(out["text_loss"]), layers={k: v.float().cpu().tolist() for k, v in out["layer_losses"].items()})
    hip_states = {l: t.detach().float().cpu() for l, t in out["layer_states"].items()}         # what the HIP heads actually read (bf16 values)
    gn = float(eng.ps.grad.norm())
    assert gn > 0 and gn == gn
    Wc = {k: v.detach().cpu() for k, v in W.items()}
    del eng, out, W
    torch.cuda.empty_cache()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ocfg = O.make_config(**{k: v for k, v in cfg.to_dict().items() if k in vars(O.make_config())})
    t0 = time.time()
    with torch.no_grad():
        ref = O.forward(Wc, batch, ocfg, need_logits=False)
    print(f"[parity] config1: bf16 CPU oracle forward {time.time() - t0:.1f} s on {torch.get_num_threads()} threads; "
          f"HIP loss {mine['loss']:.5f} vs {float(ref['loss']):.5f}")
    check("config1_vs_bf16_cpu_path/loss_rel", rel(mine["loss"], ref["loss"]), 1e-3)
    check("config1_vs_bf16_cpu_path/text_loss_rel", rel(mine["text"], ref["text_loss"]), 1e-3)
    assert sorted(ref["layer_losses"]) == sorted(mine["layers"]) == [("depth", 17), ("gen", 19), ("seg", 17)]
    # ---- which side is nearer the truth at FULL size?  The heads + _emb_loss of every distillation layer are re-run on the CPU
    # from the HIP step's OWN tapped layer states (so the decoder's bf16 noise is common to all three): (a) fp32 arithmetic on the bf16-rounded
    # weights = truth, (b) the reference-style bf16 path (bf16 weights / activations / normalisation / logits: ola_utils.py:114-115,
    # base_ola_vlm.py:306-316).  The HIP terms (fp32 normalisation, DESIGN deviation ii) are held to max(1e-3, 1.5 x the bf16 path's own deviation
    # from truth) — measured, they meet north_star's 1e-3 outright, so that is the bound; the comparison with the full bf16 CPU forward stays as a
    # cross-check at the old bounds (it measures the REFERENCE path's deviation: see the INFO lines).
    from parity import log
    spec = {"depth": ("image_depth", "depth_layer_indices", "depth_logit_scale"), "seg": ("image_seg", "seg_layer_indices", "seg_logit_scale"),
            "gen": ("image_gen", "img_layer_indices", "gen_logit_scale")}

    def heads_from(states, dtype):
        Wd = {k: (v.to(dtype) if v.dim() > 0 else v.float()) for k, v in Wc.items()}
        res = {}
        with torch.no_grad():
            for task, (cname, ikey, sname) in spec.items():
                for i, idx in enumerate(O.layer_indices(getattr(ocfg, cname)[ikey])):
                    pred, _ = O.head_forward(states[idx].to(dtype), task, i, Wd, ocfg)
                    res[(task, idx)] = [float(x) for x in O.emb_loss(pred, batch[f"{task}_mask"].float(), batch[f"{task}_target"].to(dtype), Wd.get(sname),
                                                                      ocfg.contrastive_loss_weight)]
        return res
    t0 = time.time()
    truth = heads_from(hip_states, torch.float32)
    refb = heads_from(hip_states, BF)
    print(f"[parity] config1: heads + _emb_loss from the HIP layer states, fp32 and bf16 on the CPU: {time.time() - t0:.1f} s")
    for key, trip in ref["layer_losses"].items():
        for j, (nm, old_bound) in enumerate((("emb", 1e-2), ("sl1", 1e-3), ("con", 1e-2))):
            hip_dev, ref_dev = rel(mine["layers"][key][j], truth[key][j]), rel(refb[key][j], truth[key][j])
            log(f"config1_heads_from_hip_states/{key[0]}@{key[1]}/{nm}_INFO_bf16_cpu_path_vs_fp32_truth", ref_dev)
            # measured (round 5, B = 8, D up to 884 736): HIP vs truth <= 1.6e-4 on every term (depth / seg: ~1e-6) while the bf16 CPU path sits
            # 2.3e-3 .. 5.0e-3 from the same truth: the 5e-3 "gap" of the line below is the reference path's own bf16 normalisation
            check(f"config1_heads_from_hip_states/{key[0]}@{key[1]}/{nm}_hip_vs_fp32_truth", hip_dev, 1e-3)
            check(f"config1_vs_bf16_cpu_path/layer_loss/{key[0]}@{key[1]}/{nm}_rel", rel(mine["layers"][key][j], trip[j]), old_bound)
