"""Host logic of the splice (visper_lm_amd/splice.py) against the oracle's restatement of prepare_inputs_labels_for_multimodal
(ola_arch.py:256-444), bit-exact (integer / row-copy work): the gather tables applied with numpy must reproduce the oracle's
inputs_embeds, labels, attention_mask and position_ids — right and left padding, ragged batches, a text-only sample, two images
in one sample, truncation — and the backward tables must be the exact inverses of the forward gather."""
import numpy as np
import pytest
import torch

from oracle import visper_oracle as O
from visper_lm_amd import splice
from visper_lm_amd.config import VisperConfig

H = 16


def _case(side="right", ragged=False, no_image=False, two_images=False, max_len=4096, aux="gen-depth-seg", nt=8, layout="pooled"):
    cfg = VisperConfig(vocab_size=500, hidden_size=H, num_hidden_layers=2, aux_mode=aux, num_task_tokens=nt,
                       tokenizer_padding_side=side, tokenizer_model_max_length=max_len, task_token_layout=layout)
    ocfg = O.make_config(vocab_size=500, hidden_size=H, num_hidden_layers=2, aux_mode=aux, num_task_tokens=nt,
                         tokenizer_padding_side=side, tokenizer_model_max_length=max_len, task_token_layout=layout)
    g = torch.Generator().manual_seed(3)
    B, T = 3, 61
    ids = torch.randint(0, 500, (B, T), generator=g)
    ids[:, 26] = O.IMAGE_TOKEN_INDEX
    if no_image:
        ids[1, 26] = 9
    if two_images:
        ids[2, 40] = O.IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, :30] = O.IGNORE_INDEX
    am = torch.ones(B, T, dtype=torch.bool)
    if ragged:
        am[1, 44:] = False
        am[2, 57:] = False
    n_img = B + (1 if two_images else 0)
    W = {"model.embed_tokens.weight": torch.randn(500, H, generator=g)}
    for t in aux.split("-") if aux else []:
        W[f"model.special_{t}_tokens"] = torch.randn(8 if t == "gen" else 576, H, generator=g)
    feats = torch.randn(n_img, 576, H, generator=g)
    return cfg, ocfg, ids, am, labels, W, feats


def _apply(plan, W, feats, ocfg):
    """inputs_embeds from the tables, exactly what vp_gather_rows does on the device."""
    t = plan["tables"]
    tok = torch.cat(O.task_token_rows(W, ocfg), 0) if plan["n_tok_rows"] else torch.zeros(1, H)
    srcs = [W["model.embed_tokens.weight"], feats.reshape(-1, H), tok]
    x = torch.zeros(plan["B"] * plan["S"], H)
    for k in (0, 1, 2):
        m = torch.from_numpy(t["kind"] == k)
        x[m] = srcs[k][torch.from_numpy(t["row"][t["kind"] == k].astype(np.int64))]
    if "present" in t:
        p = torch.from_numpy(t["present"].astype(np.int64))
        x = torch.where((p >= 0)[:, None], x[p.clamp_min(0)], torch.zeros(1, H))
    return x.view(plan["B"], plan["S"], H)


@pytest.mark.parametrize("side", ["right", "left"])
@pytest.mark.parametrize("kw", [dict(), dict(ragged=True), dict(ragged=True, no_image=True), dict(two_images=True, ragged=True),
                                dict(max_len=640), dict(aux="", nt=0, ragged=True),
                                # IFT-stage "emb" layout (llava_arch.py:259-260): raw 576 + 576 + 8 task-token rows behind every image
                                dict(layout="raw", ragged=True), dict(layout="raw", two_images=True, ragged=True, no_image=True),
                                dict(layout="raw", aux="seg-gen", max_len=1300)])
def test_plan_reproduces_oracle_splice(side, kw):
    cfg, ocfg, ids, am, labels, W, feats = _case(side=side, **kw)
    plan = splice.host_plan(cfg, [], ids.numpy(), am.numpy(), labels.numpy())
    pid, oam, emb, olab = O.prepare_inputs_labels_for_multimodal(ids, am, labels, feats, W, ocfg)
    assert plan["S"] == emb.shape[1]
    assert np.array_equal(plan["labels"], olab.numpy())
    assert np.array_equal(plan["attention_mask"], oam.numpy())
    assert np.array_equal(plan["position_ids"], pid.numpy())
    assert torch.equal(_apply(plan, W, feats, ocfg), emb)
    # shifted labels are laid out for the kernels (left-aligned rows): same multiset of supervised targets as the reference's shift
    want = olab[:, 1:][oam[:, 1:] & oam[:, :-1]] if side == "left" else olab[:, 1:].reshape(-1)
    got = plan["shift_labels"]
    assert sorted(got[got != O.IGNORE_INDEX].tolist()) == sorted(want[want != O.IGNORE_INDEX].tolist())
    assert plan["n_valid"] == int((got != O.IGNORE_INDEX).sum())
    # labelled-row tables of the lm_head / cross-entropy compaction: exactly the rows with a label, and their inverse
    t = plan["tables"]
    assert np.array_equal(t["ce_rows"], np.flatnonzero(got != O.IGNORE_INDEX)) and np.array_equal(plan["ce_labels"], got[t["ce_rows"]])
    assert np.array_equal(np.flatnonzero(t["ce_inv"] >= 0), t["ce_rows"]) and np.array_equal(t["ce_inv"][t["ce_rows"]], np.arange(t["ce_rows"].size))
    assert np.array_equal(t["ce_inv_kind"], np.where(t["ce_inv"] >= 0, 0, -1)) and not t["ce_kind"].any()


@pytest.mark.parametrize("side", ["right", "left"])
def test_plan_with_image_groups_reproduces_oracle_flat_merge(side):
    """list / 5-D `images` (ola_arch.py:262-275, mm_patch_merge_type "flat"): entry j holds n_j images whose features are flattened to n_j * 576
    rows that replace ONE <image> token (followed by the task tokens).  Entries of 2, 1 (text-only sample: consumed, unused), 1 and 3 images."""
    cfg, ocfg, ids, am, labels, W, _ = _case(side=side, ragged=True, no_image=True, two_images=True)
    gs = [2, 1, 1, 3]
    g = torch.Generator().manual_seed(11)
    groups = [torch.randn(n * 576, H, generator=g) for n in gs]
    plan = splice.host_plan(cfg, [], ids.numpy(), am.numpy(), labels.numpy(), group_sizes=gs)
    pid, oam, emb, olab = O.prepare_inputs_labels_for_multimodal(ids, am, labels, groups, W, ocfg)
    assert plan["S"] == emb.shape[1] == (57 - 2) + (1 + 3) * 576 + 2 * 24 and plan["n_feat"] == 7 * 576 and plan["n_img"] == 4
    assert np.array_equal(plan["labels"], olab.numpy()) and np.array_equal(plan["attention_mask"], oam.numpy())
    assert np.array_equal(plan["position_ids"], pid.numpy())
    assert torch.equal(_apply(plan, W, torch.cat(groups, 0), ocfg), emb)
    t = plan["tables"]
    for r, dst in enumerate(t["img_dst"]):                        # backward table = inverse of the gather, unused group (text-only sample) -> -1
        if dst >= 0:
            assert t["kind"][dst] == 1 and t["row"][dst] == r
    assert (t["img_dst"] >= 0).sum() == (t["kind"] == 1).sum() == 6 * 576 and (t["img_dst"][2 * 576:3 * 576] == -1).all()
    with pytest.raises(IndexError):
        splice.host_plan(cfg, [], ids.numpy(), am.numpy(), labels.numpy(), group_sizes=[2, 1, 1])       # fewer entries than the batch consumes


def test_backward_tables_are_the_inverse_of_the_gather():
    cfg, ocfg, ids, am, labels, W, feats = _case(ragged=True, two_images=True)
    plan = splice.host_plan(cfg, [], ids.numpy(), am.numpy(), labels.numpy())
    t = plan["tables"]
    M = plan["B"] * plan["S"]
    # every image row that survived lands exactly where kind == 1 reads it
    for r, dst in enumerate(t["img_dst"]):
        if dst >= 0:
            assert t["kind"][dst] == 1 and t["row"][dst] == r
    assert (t["img_dst"] >= 0).sum() == (t["kind"] == 1).sum()
    ts = t["tok_src"].reshape(plan["n_tok_rows"], plan["tok_cnt"])
    for j in range(ts.shape[0]):
        for i in range(ts.shape[1]):
            if ts[j, i] >= 0:
                assert t["kind"][ts[j, i]] == 2 and t["row"][ts[j, i]] == j
    assert (ts >= 0).sum() == (t["kind"] == 2).sum()                  # two images in one sample: both task-token blocks feed the gradient
    assert np.array_equal(t["embed_idx"] >= 0, t["kind"] == 0) and M == t["kind"].size


def test_head_tables_match_forward_emb_predictor_selection():
    """base_ola_vlm.py:413-441 via the oracle's head_inputs: the row tables select the same state rows, and the inverse table
    routes every selected row of every head back to its state row."""
    cfg, ocfg, ids, am, labels, W, feats = _case()
    tasks = [("depth", 0, 1), ("seg", 0, 0), ("seg", 1, 1), ("gen", 0, 1)]
    plan = splice.host_plan(cfg, tasks, ids.numpy(), am.numpy(), labels.numpy())
    B, S = plan["B"], plan["S"]
    state = torch.arange(B * S, dtype=torch.float32).view(B, S, 1).expand(B, S, 2).contiguous()
    for task in ("depth", "seg", "gen"):
        x, lat = O.head_inputs(state, task, W, ocfg)
        h = plan["heads"][task]
        assert np.array_equal(x[..., 0].reshape(-1).numpy().astype(np.int64), h["rows_host"].astype(np.int64))
        if task == "gen":
            assert np.array_equal(lat[0, :, 0].numpy().astype(np.int64), h["sel"][h["lat_x"]].astype(np.int64))
    inv = plan["tables"]["inv:1"].reshape(B * S, 3)                   # layer 1 is read by depth, seg(1), gen — in Engine.tasks order
    off = 0
    for j, task in enumerate(("depth", "seg", "gen")):
        rows = plan["heads"][task]["rows_host"]
        assert np.array_equal(inv[rows, j], off + np.arange(rows.size))
        assert (inv[:, j] >= 0).sum() == rows.size
        off += rows.size


def test_task_token_rows_per_layout():
    from visper_lm_amd.config import task_token_rows
    cfg, *_ = _case()
    assert task_token_rows(cfg) == [("gen", 8, False), ("depth", 8, True), ("seg", 8, True)]
    cfg, ocfg, ids, am, labels, W, feats = _case(layout="raw")
    assert task_token_rows(cfg) == [("gen", 8, False), ("depth", 576, False), ("seg", 576, False)]
    plan = splice.host_plan(cfg, [], ids.numpy(), am.numpy(), labels.numpy())
    assert plan["n_tok_rows"] == 8 + 576 + 576 and plan["S"] == 61 - 1 + 576 + plan["n_tok_rows"]
    assert task_token_rows(_case(aux="", nt=0)[0]) == []
    with pytest.raises(NotImplementedError):                         # heads slice by num_task_tokens-row blocks: pooled layout only
        splice.host_plan(cfg, [("seg", 0, 1)], ids.numpy(), am.numpy(), labels.numpy())


def test_head_tables_without_task_tokens_select_the_whole_state():
    """num_task_tokens == 0 (base_ola_vlm.py:420-422, 429-430): every head reads the whole layer state (pass_text_to_aux) and brings its
    own latents: lat_rep tiles the (num_queries, dim) parameter over the batch, lat_bwd is the transpose of that tiling."""
    cfg, ocfg, ids, am, labels, W, feats = _case(aux="gen-depth-seg", nt=0)
    tasks = [("depth", 0, 1), ("seg", 0, 0), ("gen", 0, 1)]
    plan = splice.host_plan(cfg, tasks, ids.numpy(), am.numpy(), labels.numpy())
    B, S = plan["B"], plan["S"]
    assert S == 61 - 1 + 576 and plan["n_tok_rows"] == 0
    state = torch.arange(B * S, dtype=torch.float32).view(B, S, 1).expand(B, S, 2).contiguous()
    for task, nq in (("depth", 576), ("seg", 576), ("gen", 1)):
        x, lat = O.head_inputs(state, task, {}, ocfg)
        h = plan["heads"][task]
        assert lat is None and h["mode"] == "own" and h["nq"] == nq and h["n_x"] == S
        assert np.array_equal(x[..., 0].reshape(-1).numpy().astype(np.int64), h["rows_host"].astype(np.int64))
        assert np.array_equal(h["xin_row"], h["rows_host"]) and not h["xin_kind"].any()
        assert np.array_equal(h["lat_rep"], np.tile(np.arange(nq), B))
        lb = h["lat_bwd"].reshape(nq, B)
        assert h["lat_cnt"] == B and np.array_equal(lb, np.arange(B)[None, :] * nq + np.arange(nq)[:, None])


def test_ragged_left_padding_with_heads_is_refused():
    cfg, ocfg, ids, am, labels, W, feats = _case(side="left", ragged=True)
    with pytest.raises(NotImplementedError):
        splice.host_plan(cfg, [("seg", 0, 1)], ids.numpy(), am.numpy(), labels.numpy())
    cfg.tokenizer_padding_side = "sideways"
    with pytest.raises(ValueError):
        splice.host_plan(cfg, [], ids.numpy(), am.numpy(), labels.numpy())
