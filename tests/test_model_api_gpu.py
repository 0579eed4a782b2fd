"""Drop-in API mirror on the GPU: reference-named classes, state-dict round trip, `.loss.backward()` semantics."""
import json

import numpy as np
import pytest
import torch

from parity import check, log, max_rel, scalar_grad_yardstick, scalar_grad_bound

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases
    from visper_lm_amd.model import OlaLlavaLlamaForCausalLM, OlaLlavaLlamaConfig
    ocfg, W, batch, g = cases.tiny_llama_case()
    cfg = OlaLlavaLlamaConfig(**vars(ocfg))
    model = OlaLlavaLlamaForCausalLM(cfg, init="empty")
    # the golden fixture's names use the flattened (transformers 5.x) CLIP prefix; the mirror uses the 4.41.1 (pinned) nesting
    sd = {k.replace("model.vision_tower.vision_tower.", "model.vision_tower.vision_tower.vision_model.")
          if k.startswith("model.vision_tower.vision_tower.") else k: v for k, v in W.items()}
    missing = model.load_state_dict(sd, strict=True)
    model.reload_frozen()
    return model, cfg, batch, g


def test_reference_surface(setup):
    model, cfg, batch, g = setup
    assert model.NUM_SYS_TOKENS == 38 and model.get_model() is model.model
    assert model.get_vision_tower() is model.model.vision_tower
    assert model.model.mm_projector[0].weight.shape == (cfg.hidden_size, cfg.mm_hidden_size)
    assert model.token_order == ["gen", "depth", "seg"] and model.num_task_tokens == 8
    assert model.depth_tokens.shape == (576, cfg.hidden_size) and model.gen_tokens.shape == (8, cfg.hidden_size)
    assert model.seg_layer_indices == [1, 2] and model.img_gen_loss_weight == 0.5
    tr = {n for n, p in model.named_parameters() if p.requires_grad}
    assert tr == set(json.loads(str(g["trainable"])))


def test_forward_backward_like_reference(setup):
    model, cfg, batch, g = setup
    B = batch["input_ids"].shape[0]
    out = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"],
                images=batch["images"].cuda(), gen_mask=torch.ones(B).cuda(), seg_mask=torch.ones(B).cuda(), depth_mask=torch.ones(B).cuda(),
                gen_target=batch["gen_target"].cuda(), depth_target=batch["depth_target"].cuda(), seg_target=batch["seg_target"].cuda())
    check("model_api/loss_rel_vs_reference_golden", abs(float(out.loss) - float(g["keep_loss"])) / float(g["keep_loss"]), 1e-3)
    assert len(out.seg_embs) == 2 and len(out.image_embs) == 1 and len(out.depth_embs) == 1
    out.loss.backward()
    none_ref = set(json.loads(str(g["keep_grad_none"])))
    from oracle import cases, visper_oracle as O
    ocfg, W, _, _ = cases.tiny_llama_case()
    scal = [n for n, p in model.named_parameters() if p.requires_grad and p.numel() == 1 and n not in none_ref]
    yard = scalar_grad_yardstick(O, ocfg, W, batch, scal)               # logit scales: absolute bound against the bf16 CPU path's own deviation
    for n, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        ref = float(g[f"keep_gradnorm::{n}"]) if n not in none_ref else 0.0
        got = float(p.grad.float().norm())
        if ref == 0.0:
            assert got == 0.0, n
        elif n in yard:
            log(f"model_api/gradnorm/{n}_INFO_bf16_cpu_path_abs_dev", yard[n][1])
            check(f"model_api/gradnorm/{n}_abs_vs_reference_golden", abs(got - ref), scalar_grad_bound(ref, yard[n][1], 0.2))
        else:
            check(f"model_api/gradnorm/{n}_rel_vs_reference_golden", abs(got - ref) / ref, 2.5e-2)


def test_forward_returns_what_the_reference_returns(setup):
    """ola_llama.py:113-122,170-188: fp32 logits [B, S, V] always (also without labels), every decoder-layer state (output_hidden_states=True
    is forced), the reference's tuple for return_dict=False; config.reference_outputs=False is the lean opt-in."""
    model, cfg, batch, g = setup
    L = cfg.num_hidden_layers
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], images=batch["images"].cuda())
    tg = dict(gen_target=batch["gen_target"].cuda(), depth_target=batch["depth_target"].cuda(), seg_target=batch["seg_target"].cuda())
    with torch.no_grad():
        out = model(**kw)                                              # the labels=None inference call
    assert out.loss is None
    assert tuple(out.logits.shape) == tuple(int(v) for v in g["logits_shape"]) and out.logits.dtype == torch.float32
    assert len(out.hidden_states) == L + 1 == int(g["n_hidden_states"])
    ref = g["logits_sub"]
    check("model_api/nolabel_logits_sub_maxrel_vs_reference_golden",
          float(np.abs(out.logits[:, ::41, ::997].cpu().numpy() - ref).max() / np.abs(ref).max()), 3e-2)
    for li in (0, 2, 3, 4):
        ref = g[f"hidden{li}_sub"]
        got = out.hidden_states[li].float().cpu()[:, ::13, ::3].numpy()
        check(f"model_api/hidden_states[{li}]_sub_maxrel_vs_reference_golden", float(np.abs(got - ref).max() / np.abs(ref).max()), 3e-2)
    out2 = model(labels=batch["labels"], **kw, **tg)                   # with labels: the same fields + loss
    # round 6: a training call hands the logits out LAZILY (frozen lm_head): nothing is computed until the field is read; out[0] / "loss" / keys()
    # do not touch it; the tensor read is bit-identical to the label-less call's (and stays materialised afterwards)
    from visper_lm_amd.model.language_model import _Lazy
    assert isinstance(object.__getattribute__(out2, "logits"), _Lazy)
    assert out2[0] is out2.loss and "logits" in out2 and "logits" in out2.keys() and out2["loss"] is out2.loss
    assert isinstance(object.__getattribute__(out2, "logits"), _Lazy)
    assert tuple(out2.logits.shape) == tuple(out.logits.shape) and len(out2.hidden_states) == L + 1
    assert torch.is_tensor(object.__getattribute__(out2, "logits")) and out2.logits.dtype == torch.float32
    assert torch.equal(out2.logits, out.logits) and torch.equal(out2[1], out.logits)
    eager = model(labels=batch["labels"], output_logits=True, **kw, **tg)           # computed inside the step instead: same bits
    assert torch.is_tensor(object.__getattribute__(eager, "logits")) and torch.equal(eager.logits, out.logits)
    check("model_api/loss_rel_vs_reference_golden(reference_outputs)", abs(float(out2.loss) - float(g["keep_loss"])) / float(g["keep_loss"]), 1e-3)
    tup = model(labels=batch["labels"], return_dict=False, **kw, **tg)
    assert isinstance(tup, tuple) and len(tup) == 3 and tup[0].shape == () and torch.equal(tup[1], out.logits) and len(tup[2]) == L + 1
    cfg.reference_outputs = False                                      # lean mode: labelled rows only through lm_head + CE
    try:
        lean = model(labels=batch["labels"], **kw, **tg)
        assert lean.logits is None and len(lean.hidden_states) == 2
        assert abs(float(lean.loss) - float(out2.loss)) <= 1e-6 * abs(float(out2.loss))
        assert len(model(labels=batch["labels"], output_hidden_states=True, **kw, **tg).hidden_states) == L + 1
        with torch.no_grad():
            assert model(**kw).logits is not None                      # no labels: logits are the only product of the call
    finally:
        cfg.reference_outputs = True


def test_encode_images_and_prepare_inputs(setup):
    model, cfg, batch, g = setup
    feats = model.encode_images(batch["images"].cuda())
    assert feats.shape == (2, 576, cfg.hidden_size)
    r = model.prepare_inputs_labels_for_multimodal(batch["input_ids"], None, batch["attention_mask"], None, batch["labels"],
                                                   batch["images"].cuda())
    assert r[0] is None and r[4].shape == (2, 658, cfg.hidden_size) and r[5].shape == (2, 658)
    ref = g["hidden0_sub"]
    got = r[4].float().cpu()[:, ::13, ::3].numpy()
    check("model_api/inputs_embeds_sub_maxrel_vs_reference_golden", float(np.abs(got - ref).max() / np.abs(ref).max()), 2e-2)


def test_mm_projector_checkpoint_round_trip(tmp_path):
    """f-4: `mm_projector.bin` in the reference's format (llava_trainer.py:1006-1014) written from the engine and loaded back."""
    import torch
    from oracle import cases
    from visper_lm_amd import data
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.engine import Engine
    ocfg, W, batch, g = cases.tiny_llama_case()
    eng = Engine(VisperConfig(**vars(ocfg)))
    eng.load_weights(W)
    path = str(tmp_path / "mm_projector.bin")
    data.save_mm_projector(eng, path)
    sd = torch.load(path)
    assert sorted(sd) == sorted(k for k in eng.ps.index if "mm_projector" in k) and all(v.dtype == torch.bfloat16 for v in sd.values())
    before = {k: eng.ps.p(k).clone() for k in sd}
    for k in sd:
        eng.ps.p(k).zero_()
    loaded = data.load_mm_projector(eng, path)
    assert sorted(loaded) == sorted(sd)
    for k in sd:
        assert torch.equal(eng.ps.p(k), before[k].to(torch.bfloat16).float())
        assert torch.equal(eng.ps.w(k), before[k].to(torch.bfloat16))
    # the reference's loader also accepts keys without the leading "model." (builder.py:131-137)
    torch.save({k[len("model."):]: v for k, v in sd.items()}, path)
    assert sorted(data.load_mm_projector(eng, path)) == sorted(sd)


def test_attached_gpu_teachers_feed_the_step():
    """f-3 through the drop-in API: with teachers attached, `<task>_pixels` tensors replace precomputed `<task>_target`s and the loss equals
    the one obtained by passing the teachers' outputs explicitly."""
    import json
    import torch
    from oracle import cases, weights as WT
    from visper_lm_amd.model import OlaLlavaLlamaForCausalLM
    from visper_lm_amd.model.language_model import OlaLlavaLlamaConfig
    from visper_lm_amd.teachers import DinoV2DepthTeacher
    ocfg, W, batch, g = cases.tiny_llama_case()
    gd = cases.load_golden("dinov2_teacher.npz")
    dims = json.loads(str(gd["dims"]))
    # a 1024-wide DINOv2 of 2 blocks so that its output matches the depth head's target width
    sh = DinoV2DepthTeacher.shapes(1024, 2)
    t = DinoV2DepthTeacher(1024, 2, 16, taps=(0, 1))
    t.load_weights({k: WT.param(k, s) for k, s in sh.items()})
    model = OlaLlavaLlamaForCausalLM(OlaLlavaLlamaConfig(**vars(ocfg)))
    model.load_state_dict({k: v for k, v in W.items() if k in model.state_dict()}, strict=False)
    model.reload_frozen()
    model.attach_teachers(depth=t)
    px = WT.tensor("teacher_px", (2, 3, 336, 336)).cuda()
    common = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"].cuda(),
                  gen_target=batch["gen_target"].cuda(), seg_target=batch["seg_target"].cuda())
    out1 = model(**common, depth_pixels=px)
    out2 = model(**common, depth_target=t.forward(px))
    assert torch.isfinite(out1.loss) and torch.equal(out1.loss.detach(), out2.loss.detach())


def _mirror_from_tiny():
    from oracle import cases
    from visper_lm_amd.model import OlaLlavaLlamaForCausalLM, OlaLlavaLlamaConfig
    ocfg, W, batch, g = cases.tiny_llama_case()
    model = OlaLlavaLlamaForCausalLM(OlaLlavaLlamaConfig(**vars(ocfg)), init="empty")
    sd = {k.replace("model.vision_tower.vision_tower.", "model.vision_tower.vision_tower.vision_model.")
          if k.startswith("model.vision_tower.vision_tower.") else k: v for k, v in W.items()}
    model.load_state_dict(sd, strict=True)
    model.reload_frozen()
    B = batch["input_ids"].shape[0]
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"].cuda(),
              gen_mask=torch.ones(B).cuda(), seg_mask=torch.ones(B).cuda(), depth_mask=torch.ones(B).cuda(),
              gen_target=batch["gen_target"].cuda(), depth_target=batch["depth_target"].cuda(), seg_target=batch["seg_target"].cuda())
    return model, kw


def test_model_api_with_engine_optimizer_makes_progress():
    """model(**batch).loss.backward() + Engine.optimizer_step must train — the nn.Parameters and the engine's
    flat master are ONE state: the loss falls, state_dict() moves, and a second model built from that state_dict reproduces the loss."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    model, kw = _mirror_from_tiny()
    before = {k: v.clone() for k, v in model.state_dict().items() if "mm_projector" in k or "_heads.0.projector.proj_in" in k}
    losses = []
    for i in range(4):
        out = model(**kw)
        out.loss.backward()
        losses.append(float(out.loss))
        if i % 2 == 0:
            model.optimizer_step(lr=1e-3)                        # convenience wrapper ...
        else:
            model._get_engine().optimizer_step(lr=1e-3)          # ... and the INTEGRATION.md recipe (engine stepped behind the model's back)
    assert losses[-1] < losses[0] - 1e-3 and all(b < a for a, b in zip(losses, losses[1:])), losses
    sd = model.state_dict()
    assert all(not torch.equal(sd[k], v) for k, v in before.items())
    eng = model._get_engine()
    for n in ("model.mm_projector.0.weight", "image_seg_heads.0.projector.proj_in.weight"):
        assert torch.equal(sd[n], eng.ps.p(n).to(sd[n].dtype).reshape(sd[n].shape))
    final = float(model(**kw).loss)
    from visper_lm_amd.model import OlaLlavaLlamaForCausalLM
    clone = OlaLlavaLlamaForCausalLM(model.config, init="empty")
    clone.load_state_dict(sd, strict=True)
    clone.reload_frozen()
    assert abs(float(clone(**kw).loss) - final) < 2e-2 * abs(final)          # bf16 state_dict vs the fp32 master it was cast from


def test_model_api_with_external_torch_optimizer():
    """HF Trainer's shape (ola_vlm_train.py:1297-1309): torch.optim.AdamW over model.parameters() steps the nn.Parameters; the
    engine picks the new values up through their version counters."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    model, kw = _mirror_from_tiny()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.0)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        out = model(**kw)
        out.loss.backward()
        opt.step()
        losses.append(float(out.loss))
    assert losses[2] < losses[1] < losses[0], losses
    eng = model._get_engine()
    model._sync_trainable()
    p = dict(model.named_parameters())["model.mm_projector.2.weight"]
    assert torch.equal(eng.ps.w("model.mm_projector.2.weight"), p.detach().to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ trainer-shaped loops
def _instances(batch):
    """Dataset items exactly as LazySupervisedDataset yields them (ola_vlm_train.py:774-880): 1-D ids / labels, one image tensor, the
    PIL image (here: its index — the teachers are replaced by synthetic targets) and the per-sample task masks."""
    n = batch["input_ids"].shape[0]
    return [dict(input_ids=batch["input_ids"][i], labels=batch["labels"][i], image=batch["images"][i], pil_image=i,
                 seg_mask=1, depth_mask=1, gen_mask=1) for i in range(n)]


def _teacher_hooks(model, batch):
    """Frozen-teacher features as the reference's hooks return them (base_ola_vlm.py:323-397), looked up by the 'PIL image' index."""
    dev = "cuda"
    model._get_gen_feats = lambda pil, device: torch.stack([batch["gen_target"][i] for i in pil]).to(dev)
    model._get_dav2_feats = lambda pil, device: ([(torch.stack([batch["depth_target"][i] for i in pil]).to(dev), None)], None)
    model._get_seg_targets = lambda pil, preds: torch.stack([batch["seg_target"][i] for i in pil]).to(dev)


def _oracle_adamw_curve(ocfg, W, batch, trainable, steps, lr):
    """The reference's training semantics on the CPU: autograd through the oracle + torch.optim.AdamW (HF `adamw_torch`, weight decay
    0: pretrain.sh) on fp32 MASTER copies of the trainable parameters, with the forward run on their bf16 rounding (straight-through) —
    DeepSpeed's bf16 mode, which the reference trains in (scripts/zero2.json "bf16": auto): a step smaller than half a bf16 ulp of a
    weight does not reach the forward until the master has drifted far enough, exactly as on the engine's master/shadow pair."""
    from oracle import visper_oracle as O
    BF = torch.bfloat16
    Wq = {k: v.to(BF).float() for k, v in W.items()}
    master = {k: Wq[k].clone().requires_grad_(True) for k in trainable}
    bq = {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in batch.items()}
    opt = torch.optim.AdamW(list(master.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        Wf = dict(Wq)
        for k, p in master.items():
            Wf[k] = p + (p.detach().to(BF).float() - p.detach()) if p.dim() > 0 else p       # logit scales stay fp32 (reference: fp32 Parameter)
        out = O.forward(Wf, bq, ocfg)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    return losses


def test_collator_to_model_to_engine_optimizer_pt_stage_matches_oracle_adamw():
    """what HF Trainer does per step (ola_vlm_train.py:1297-1327: collator -> model(**batch) -> loss.backward() ->
    optimizer.step()) driven through the drop-in class for 3 steps, against the oracle stepping torch.optim.AdamW."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import cases
    from visper_lm_amd import data
    ocfg, W, batch, g = cases.tiny_llama_case()
    model, _ = _mirror_from_tiny()
    _teacher_hooks(model, batch)
    coll = data.Collator(pad_token_id=0, model_max_length=4096)
    lr, steps = 1e-3, 3
    losses = []
    for _ in range(steps):
        b = coll(_instances(batch))
        out = model(**{k: (v.cuda() if k == "images" else v) for k, v in b.items()})
        out["loss"].backward()
        model.optimizer_step(lr=lr)
        losses.append(float(out.loss))
    ref = _oracle_adamw_curve(ocfg, W, batch, json.loads(str(g["trainable"])), steps, lr)
    for i, (a, r) in enumerate(zip(losses, ref)):
        check(f"trainer_loop_pt/step{i}_loss_rel_vs_oracle_adamw", abs(a - r) / abs(r), 5e-4)
    check("trainer_loop_pt/loss_drop_rel_dev", abs((losses[0] - losses[-1]) - (ref[0] - ref[-1])) / abs(ref[0] - ref[-1]), 1e-2)
    assert losses[-1] < losses[0]


def test_llava_llama_ift_mirror_drop_in():
    """VERDICT r1 #1 / north_star: the LlavaMetaForCausalLM / LlavaLlamaForCausalLM surface (llava_arch.py:210,295-298;
    llava_llama.py:51,73-119).  Reference golden (its own LlavaLlamaForCausalLM): loss and every parameter-gradient norm through
    `.loss.backward()`; the Parameters ARE the engine's flat store (no second copy of the LLM); 3 collator -> model -> optimizer
    steps follow the oracle's AdamW curve; sharded safetensors save_pretrained / from_pretrained round trip."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import tempfile
    from oracle import cases
    from visper_lm_amd import data
    from visper_lm_amd.model import LlavaLlamaForCausalLM, LlavaConfig, LlavaMetaForCausalLM, CausalLMOutputWithPast
    ocfg, W, batch, g = cases.tiny_ift_case()
    cfg = LlavaConfig(**vars(ocfg))
    assert cfg.train_llm and not cfg.aux_heads and cfg.model_type == "llava_llama"
    model = LlavaLlamaForCausalLM(cfg, init="empty")
    assert isinstance(model, LlavaMetaForCausalLM)
    sd = {k.replace("model.vision_tower.vision_tower.", "model.vision_tower.vision_tower.vision_model.")
          if k.startswith("model.vision_tower.vision_tower.") else k: v for k, v in W.items()}
    model.load_state_dict(sd, strict=True)
    tr = json.loads(str(g["trainable"]))
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == tr
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"].cuda())
    out = model(**kw)
    assert isinstance(out, CausalLMOutputWithPast) and out["loss"] is out.loss and out[0] is out.loss
    check("ift_mirror/loss_rel_vs_reference_golden", abs(float(out.loss) - float(g["loss"])) / float(g["loss"]), 1e-3)
    out.loss.backward()
    named = dict(model.named_parameters())
    for k in tr:
        ref = float(g[f"gradnorm::{k}"])
        got = float(named[k].grad.float().norm())
        if ref == 0.0:
            assert got < 1e-7, k
        else:
            check(f"ift_mirror/gradnorm/{k}_rel_vs_reference_golden", abs(got - ref) / ref, 2.5e-2)
    eng = model._get_engine()
    for k in ("model.layers.0.self_attn.q_proj.weight", "lm_head.weight", "model.embed_tokens.weight", "model.mm_projector.0.bias"):
        assert named[k].data_ptr() == eng.ps.w(k).data_ptr(), k                  # Parameter == view of the bf16 shadow
    feats = model.encode_images(batch["images"].cuda())
    assert feats.shape == (2, 576, cfg.hidden_size)
    r = model.prepare_inputs_labels_for_multimodal(batch["input_ids"], None, batch["attention_mask"], None, batch["labels"], batch["images"].cuda())
    assert r[0] is None and r[4].shape[:2] == r[5].shape
    # ---- 3 trainer-shaped steps vs the oracle's AdamW
    coll = data.Collator(pad_token_id=0, model_max_length=4096)
    inst = [dict(input_ids=batch["input_ids"][i], labels=batch["labels"][i], image=batch["images"][i]) for i in range(2)]
    lr, losses = 2e-4, []
    for _ in range(3):
        b = coll(inst)
        o = model(**{k: (v.cuda() if k == "images" else v) for k, v in b.items()})
        o.loss.backward()
        model.optimizer_step(lr=lr)
        losses.append(float(o.loss))
    ref = _oracle_adamw_curve(ocfg, W, batch, tr, 3, lr)
    for i, (a, r_) in enumerate(zip(losses, ref)):
        check(f"ift_mirror/step{i}_loss_rel_vs_oracle_adamw", abs(a - r_) / abs(r_), 3e-4)
    assert losses[-1] < losses[0]
    # ---- HF-style persistence: shards + index, bitwise state, same loss
    with tempfile.TemporaryDirectory() as d:
        model.save_pretrained(d, max_shard_size=200_000)
        import os
        assert os.path.exists(os.path.join(d, "model.safetensors.index.json")) and os.path.exists(os.path.join(d, "config.json"))
        clone = LlavaLlamaForCausalLM.from_pretrained(d)
        a, b_ = model.state_dict(), clone.state_dict()
        assert sorted(a) == sorted(b_) and all(torch.equal(a[k], b_[k]) for k in a)
        l1, l2 = float(model(**kw).loss), float(clone(**kw).loss)
        assert l1 == l2, (l1, l2)


def test_hf_trainer_drives_the_pt_mirror():
    """The reference's own driver: transformers.Trainer (LLaVATrainer's base, llava_trainer.py:217) with the supervised collator,
    its torch AdamW and gradient clipping, 3 optimizer steps on the drop-in class — the loss must fall and the trained state must be
    what the engine runs on."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import tempfile
    from oracle import cases
    from transformers import Trainer, TrainingArguments
    from visper_lm_amd import data
    ocfg, W, batch, g = cases.tiny_llama_case()
    model, kw = _mirror_from_tiny()
    _teacher_hooks(model, batch)
    l0 = float(model(**kw).loss)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            return _instances(batch)[i % 2]
    with tempfile.TemporaryDirectory() as d:
        args = TrainingArguments(output_dir=d, per_device_train_batch_size=2, max_steps=3, learning_rate=1e-3, weight_decay=0.0,
                                 lr_scheduler_type="constant", logging_steps=1, save_strategy="no", report_to=[], bf16=True,
                                 remove_unused_columns=False, dataloader_num_workers=0, dataloader_pin_memory=False, seed=0)
        trainer = Trainer(model=model, args=args, train_dataset=DS(), data_collator=data.Collator(0, 4096, pin_memory=False))
        res = trainer.train()
    assert res.global_step == 3 and torch.isfinite(torch.tensor(res.training_loss))
    l1 = float(model(**kw).loss)
    assert l1 < l0 - 1e-3, (l0, l1)
    eng = model._get_engine()
    p = dict(model.named_parameters())["model.mm_projector.0.weight"]
    assert p.data_ptr() == eng.ps.w("model.mm_projector.0.weight").data_ptr()
    model._sync_trainable()                                           # fold the Trainer's in-place steps on the bf16 Parameters into the fp32 master
    assert torch.equal(eng.ps.p("model.mm_projector.0.weight").to(torch.bfloat16).view(p.shape), p.detach())


def test_pt_checkpoint_hands_off_to_the_ift_class():
    """ADVICE r2 + VERDICT r2 missing-2/-7: the reference's PT -> IFT hand-off.  A PT-stage save_pretrained directory (config.json carries
    train_llm=false, aux_heads=true, model_type ola_llama, num_task_tokens 8, task_token_format "emb") loaded into LlavaLlamaForCausalLM:
    the whole LLM is trainable (train.py makes it so whatever the config stores), the heads are skipped, the three special_*_tokens come
    along and are spliced raw (llava_arch.py:259-260), the loss equals the IFT golden's reference loss when the weights are the golden's,
    a text-only batch (no `images`) runs (the reference's dataset attaches a zero image), and "text" is refused like the reference does."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import tempfile
    from oracle import cases
    from visper_lm_amd.model import OlaLlavaLlamaForCausalLM, OlaLlavaLlamaConfig, LlavaLlamaForCausalLM, LlavaConfig
    ocfg, W, batch, g = cases.tiny_ift_tok_case()
    pt_cfg = OlaLlavaLlamaConfig(**{k: v for k, v in vars(ocfg).items() if k not in ("aux_heads", "task_token_layout")})
    assert pt_cfg.train_llm is False and pt_cfg.aux_heads is True and pt_cfg.task_token_format == "emb"
    pt = OlaLlavaLlamaForCausalLM(pt_cfg, init="random", seed=3)
    own = dict(pt.named_parameters())
    nest = lambda k: (k.replace("model.vision_tower.vision_tower.", "model.vision_tower.vision_tower.vision_model.")
                      if k.startswith("model.vision_tower.vision_tower.") else k)
    with torch.no_grad():
        for k, v in W.items():                                       # the golden's LLM / tower / projector / token weights into the PT model
            own[nest(k)].copy_(v.to(own[nest(k)].dtype))
    with tempfile.TemporaryDirectory() as d:
        pt.save_pretrained(d)
        stored = json.load(open(f"{d}/config.json"))
        assert stored["train_llm"] is False and stored["model_type"] == "ola_llama" and stored["num_task_tokens"] == 8
        ift = LlavaLlamaForCausalLM.from_pretrained(d, strict=False)
        frozen = LlavaLlamaForCausalLM.from_pretrained(d, strict=False, freeze_llm=True)
    cfg = ift.config
    assert isinstance(cfg, LlavaConfig) and cfg.train_llm and not cfg.aux_heads and cfg.model_type == "llava_llama" and cfg.task_token_layout == "raw"
    named = dict(ift.named_parameters())
    assert not any("_heads." in k or k.endswith("logit_scale") for k in named)
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.3.mlp.down_proj.weight", "lm_head.weight", "model.embed_tokens.weight",
              "model.norm.weight", "model.mm_projector.2.weight", "model.special_depth_tokens", "model.special_seg_tokens", "model.special_gen_tokens"):
        assert named[k].requires_grad, k
    assert not any(p.requires_grad for k, p in named.items() if "vision_tower" in k)
    assert not dict(frozen.named_parameters())["model.layers.0.self_attn.q_proj.weight"].requires_grad          # explicit freeze only
    assert torch.equal(named["model.special_seg_tokens"].detach().float().cpu(), W["model.special_seg_tokens"].to(torch.bfloat16).float())
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"].cuda())
    out = ift(**kw)
    assert out.hidden_states[0].shape[1] == 58 + 576 * 3 + 8
    check("pt_to_ift/loss_rel_vs_reference_golden", abs(float(out.loss) - float(g["loss"])) / float(g["loss"]), 1e-3)
    out.loss.backward()
    for k in ("model.special_depth_tokens", "model.special_seg_tokens", "model.special_gen_tokens", "model.layers.1.mlp.up_proj.weight"):
        ref = float(g[f"gradnorm::{k}"])
        check(f"pt_to_ift/gradnorm/{k}_rel_vs_reference_golden", abs(float(named[k].grad.float().norm()) - ref) / ref, 2.5e-2)
    # text-only batch: no images, no <image> token -> a zero image per sample whose features are never read (llava_arch.py:347-354)
    ids = batch["input_ids"].clone(); ids[:, 38] = 7
    lab = ids.clone(); lab[:, :45] = -100
    ift.zero_grad(set_to_none=True)
    o2 = ift(input_ids=ids, attention_mask=batch["attention_mask"], labels=lab)
    assert o2.hidden_states[0].shape[1] == 59 and float(o2.loss) == float(o2.loss)
    o2.loss.backward()
    assert float(named["model.mm_projector.0.weight"].grad.float().abs().max()) == 0.0          # the dummy image feeds nothing
    with pytest.raises(ValueError):
        ift(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])     # <image> token without images
    with pytest.raises(ValueError, match="text"):
        LlavaLlamaForCausalLM(LlavaConfig(**{**pt_cfg.to_dict(), "task_token_format": "text"}), init="empty")
