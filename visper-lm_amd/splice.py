"""Host index plan of the multimodal splice — prepare_inputs_labels_for_multimodal's bookkeeping (ola_arch.py:256-444),
append_special_tokens' row layout (ola_arch.py:224-254) and forward_emb_predictor's token selection (base_ola_vlm.py:413-441) —
as integer tables that drive the HIP row gathers (`vp_gather_rows`, `vp_gather_sum_rows`).  Pure numpy, no device, no per-token
Python loops (one pass per sample and per <image> segment): ~1 ms for B=8, T=1449.

Padding side (ola_arch.py:408-427): the kernels always run the sequences LEFT-ALIGNED (real tokens at rows 0..len-1, keys beyond
`lens` masked, RoPE position = row index = the reference's position_ids of the real tokens in both modes).  With
tokenizer_padding_side == "left" only the PRESENTATION changes: labels / attention_mask / position_ids and the per-row outputs
(inputs_embeds, hidden, logits) are re-laid right-aligned through the `present` table.  The reference's pad rows carry no defined
values (eager attention averages all keys there, flash-attention drops them); they are zeros here."""
from __future__ import annotations

import numpy as np

from .config import IGNORE_INDEX, IMAGE_TOKEN_INDEX, task_token_rows

N_IMG_TOK = 576   # the reference hard-codes 576 image tokens in the head slicing (base_ola_vlm.py:415-418)


def head_tables(cfg, tasks, B, S):
    """forward_emb_predictor's token selection as per-task row tables into the [B*S] state (host arrays), plus the tables of each
    head's input gather [x_b ; latents_b] (TaskTokenResampler.forward, resampler.py:202-216: the latents are tiled to num_queries when
    that is a multiple of their count, else replaced by their mean) and of its transpose.
    tasks: [(task, head_i, layer_idx)] (Engine.tasks)."""
    ns, nt, order = cfg.num_sys_tokens, cfg.num_task_tokens, cfg.token_order
    out = {}
    if not tasks or S <= ns:                                   # ola_llama.py:139: heads only run when the sequence carries an image
        return out
    if nt > 0 and getattr(cfg, "task_token_layout", "pooled") != "pooled":
        raise NotImplementedError("distillation heads slice the sequence by num_task_tokens-row blocks (base_ola_vlm.py:414-417): they only "
                                  "exist with the PT stage's pooled task-token layout")
    for task in sorted({t for t, _, _ in tasks}):              # SORTED: every rank must issue the all-gathers in the same order
        k = order.index(task)
        s0 = ns + N_IMG_TOK + nt * k
        end = ns + N_IMG_TOK + nt * len(order)
        if nt == 0 or S < 600:
            sel = np.arange(S if cfg.pass_text_to_aux else min(S, ns + N_IMG_TOK), dtype=np.int32)
        else:
            parts = [np.arange(ns + N_IMG_TOK, dtype=np.int32), np.arange(s0, s0 + nt, dtype=np.int32)]
            if cfg.pass_text_to_aux:
                parts.append(np.arange(end, S, dtype=np.int32))
            sel = np.concatenate(parts)
        rows = (np.arange(B, dtype=np.int32)[:, None] * S + sel[None, :]).reshape(-1)
        if cfg.pass_text_to_aux:
            lat_x = np.arange(ns + N_IMG_TOK, ns + N_IMG_TOK + nt)               # positions of the gen latents inside x
        else:
            lat_x = np.arange(len(sel) - nt, len(sel))
        h = dict(n_x=len(sel), rows_host=rows, sel=sel, lat_x=lat_x)
        # ---- the head's input gather: kind 0 = layer-state row, kind 1 = latent source row (the (num_tokens, H) task-token parameter for
        # depth / seg; for gen the state rows of its 8 task tokens, or row b of their per-sample mean)
        hc = {"gen": cfg.image_gen, "seg": cfg.image_seg, "depth": cfg.image_depth}[task]
        n, nq = len(sel), int(hc["num_tokens"])
        if nt == 0:
            # num_task_tokens == 0: GenHead / DepthHead / OneFormerSegHead with a plain Resampler (base_ola_vlm.py:429-430, resampler.py:120-165):
            # the queries are the head's own `latents` parameter (already in the resampler's width, NOT passed through proj_in), so the input
            # gather holds the state rows only; `lat_rep` tiles the (num_queries, dim) parameter over the batch and `lat_bwd` is its transpose
            bb = np.arange(B, dtype=np.int32)[:, None]
            h.update(nq=nq, nl=nq, mode="own", xin_kind=np.zeros(B * n, np.int32), xin_row=(bb * S + sel[None, :]).reshape(-1).astype(np.int32),
                     mean_idx=None, lat_rep=np.tile(np.arange(nq, dtype=np.int32), B),
                     lat_bwd=np.ascontiguousarray((bb * nq + np.arange(nq, dtype=np.int32)[None, :]).T.reshape(-1)).astype(np.int32), lat_cnt=B)
            out[task] = h
            continue
        nl = nt if task == "gen" else nq                             # gen latents = its nt task-token rows of the state; depth / seg = the
                                                                     # (num_tokens, H) special_{task}_tokens parameter (ola_arch.py:80-90)
        mode = "same" if nl == nq else ("tile" if (nq > 1 and nl > 0 and nq % nl == 0) else "mean")
        T = n + nq
        bb = np.arange(B, dtype=np.int32)[:, None]
        kind = np.zeros((B, T), np.int32)
        rowt = np.zeros((B, T), np.int32)
        rowt[:, :n] = bb * S + sel[None, :]
        lat_state = (bb * S + sel[lat_x][None, :]).astype(np.int32) if (task == "gen" and nt > 0 and len(sel) >= nt) else None
        if task == "gen" and mode != "mean":
            rowt[:, n:] = np.tile(lat_state, (1, nq // nl))
        elif task == "gen":
            kind[:, n:] = 1
            rowt[:, n:] = bb
        else:
            kind[:, n:] = 1
            rowt[:, n:] = np.tile(np.arange(nl, dtype=np.int32), nq // nl)[None, :] if mode != "mean" else 0
        h.update(nq=nq, nl=nl, mode=mode, xin_kind=kind.reshape(-1), xin_row=rowt.reshape(-1),
                 mean_idx=None if lat_state is None else lat_state.reshape(-1))
        if task != "gen" and mode != "mean":                         # parameter row i <- sum over batch (and tile copies) of its dxin rows
            reps = nq // nl if mode == "tile" else 1
            src = bb[:, :, None] * T + n + (np.arange(reps, dtype=np.int32)[None, :, None] * nl + np.arange(nl, dtype=np.int32)[None, None, :])
            h["lat_bwd"] = np.ascontiguousarray(src.transpose(2, 0, 1).reshape(-1)).astype(np.int32)
            h["lat_cnt"] = B * reps
        out[task] = h
    return out


def inverse_tables(tasks, heads, M):
    """Per tapped layer: state row -> row of the concatenated head-input gradients, one column per head reading that layer
    (-1 = not read), in `tasks` order — the backward of the heads' row gathers as ONE gather-sum."""
    out = {}
    for l in sorted({idx for _, _, idx in tasks}):
        tl = [t for t, _, idx in tasks if idx == l]
        if not tl or not all(t in heads for t in tl):
            continue
        inv = np.full((M, len(tl)), -1, np.int32)
        off = 0
        for j, t in enumerate(tl):
            rows = heads[t]["rows_host"]
            inv[rows, j] = off + np.arange(rows.size, dtype=np.int32)
            off += rows.size
        out[l] = inv
    return out


def host_plan(cfg, tasks, ids, am=None, lab=None, group_sizes=None):
    """ids [B,T] int64 (IMAGE_TOKEN_INDEX marks an image), am [B,T] bool or None, lab [B,T] int64 or None,
    group_sizes: None (every <image> token / text-only sample consumes ONE image's 576 feature rows: a stacked 4-D `images`) or the number of
    images n_j in each entry of a list / 5-D `images` (ola_arch.py:262-275, mm_patch_merge_type "flat": entry j's features are flattened to
    n_j * 576 rows and ONE <image> token is replaced by all of them, followed by the task tokens) ->
    dict(B, S, lens_host, labels, attention_mask, position_ids, shift_labels, tables{name: int32 array}, heads, ...).

    tables: kind/row  [B*S]  gather source (0 = embed_tokens, 1 = image features, 2 = task-token rows, -1 = zeros) and row in it
            img_dst   [n_img*576]       image-feature row -> its row of [B*S] (-1: truncated away / unused slot)
            tok_src   [n_tok_rows, n_img]   task-token row j of image i -> its row of [B*S]
            embed_idx [B*S]  token id of text rows (-1 elsewhere): embed_tokens scatter-add when the LLM trains
            present(_kind) [B*S]  only for ragged left padding: presented row -> physical row
            ce_rows / ce_inv / un_rows  rows with / without a next-token label (lm_head + CE on the former, lm_head forward alone on the latter)
            rows:<task>, inv:<layer>  head gathers (head_tables / inverse_tables)."""
    side = getattr(cfg, "tokenizer_padding_side", "right")
    if side not in ("right", "left"):
        raise ValueError(f"tokenizer_padding_side={side!r}")
    B, T = ids.shape
    if am is None:
        am = np.ones((B, T), bool)
    if lab is None:
        lab = np.full((B, T), IGNORE_INDEX, np.int64)
    n_tok_rows = sum(r for _, r, _ in task_token_rows(cfg))    # rows the task tokens take behind every image (pooled or raw layout)
    tok_rows = np.arange(n_tok_rows, dtype=np.int32)
    gs = None if group_sizes is None else [int(n) for n in group_sizes]
    goff = None if gs is None else np.concatenate(([0], np.cumsum(gs))).astype(np.int64) * N_IMG_TOK      # first feature row of every entry

    def group(j):
        """(feature rows of entry j, its first row)"""
        if gs is None:
            return N_IMG_TOK, j * N_IMG_TOK
        if j >= len(gs):
            raise IndexError(f"`images` has {len(gs)} entries but the batch consumes more (one per <image> token and per text-only sample: "
                             f"ola_arch.py:347-354,372-374)")
        return gs[j] * N_IMG_TOK, int(goff[j])
    mx = cfg.tokenizer_model_max_length
    seqs = []
    img_idx = 0
    for b in range(B):
        idb, lb = ids[b][am[b]], lab[b][am[b]]
        pos = np.flatnonzero(idb == IMAGE_TOKEN_INDEX)
        if pos.size == 0:
            k, r, lo, im = np.zeros(idb.size, np.int32), idb.astype(np.int32), lb, np.full(idb.size, -1, np.int32)
            group(img_idx)                                      # (a list / 5-D `images` must have the entry: the reference indexes it)
            img_idx += 1                                        # the reference consumes one (empty) feature slot: ola_arch.py:347-354
        else:
            ks, rs, ls, ims = [], [], [], []
            bounds = np.concatenate(([-1], pos, [idb.size]))
            for i in range(bounds.size - 1):
                lo_, hi_ = int(bounds[i]) + 1, int(bounds[i + 1])
                ks.append(np.zeros(hi_ - lo_, np.int32)); rs.append(idb[lo_:hi_].astype(np.int32)); ls.append(lb[lo_:hi_])
                ims.append(np.full(hi_ - lo_, -1, np.int32))
                if i < pos.size:
                    g_rows, g0 = group(img_idx)
                    blk = g_rows + n_tok_rows
                    ks.append(np.concatenate([np.full(g_rows, 1, np.int32), np.full(n_tok_rows, 2, np.int32)]))
                    rs.append(np.arange(g0, g0 + g_rows, dtype=np.int32)); rs.append(tok_rows); ls.append(np.full(blk, IGNORE_INDEX, np.int64))
                    ims.append(np.full(blk, img_idx, np.int32))
                    img_idx += 1
            k, r, lo, im = np.concatenate(ks), np.concatenate(rs), np.concatenate(ls), np.concatenate(ims)
        if mx is not None:
            k, r, lo, im = k[:mx], r[:mx], lo[:mx], im[:mx]
        seqs.append((k, r, lo, im))
    n_img = img_idx
    lens = np.array([s[0].size for s in seqs], np.int32)
    S = int(lens.max())
    M = B * S
    kind = np.full((B, S), -1, np.int32)
    row = np.zeros((B, S), np.int32)
    imgi = np.full((B, S), -1, np.int32)
    lab2 = np.full((B, S), IGNORE_INDEX, np.int64)
    for b, (k, r, lo, im) in enumerate(seqs):
        n = k.size
        kind[b, :n], row[b, :n], lab2[b, :n], imgi[b, :n] = k, r, lo, im
    shift = np.full((B, S), IGNORE_INDEX, np.int64)            # ola_llama.py:128-131: logits[..., :-1] vs labels[..., 1:]
    shift[:, :-1] = lab2[:, 1:]
    full = bool((lens == S).all())
    fk, fr, fi = kind.reshape(-1), row.reshape(-1), imgi.reshape(-1)
    posn = np.arange(M, dtype=np.int32)
    n_feat = max(n_img, 1) * N_IMG_TOK if gs is None else max(int(goff[-1]), N_IMG_TOK)      # rows of the tower's output
    img_dst = np.full(n_feat, -1, np.int32)
    m1 = fk == 1
    img_dst[fr[m1]] = posn[m1]
    tok_src = np.full((max(n_tok_rows, 1), max(n_img, 1)), -1, np.int32)
    m2 = fk == 2
    tok_src[fr[m2], fi[m2]] = posn[m2]
    embed_idx = np.where(fk == 0, fr, -1).astype(np.int32)
    col = np.arange(S, dtype=np.int32)[None, :]
    real = col < lens[:, None]                                   # physical (left-aligned) validity
    plan = dict(B=B, S=S, n_img=n_img, n_feat=n_feat, n_valid=int((shift != IGNORE_INDEX).sum()), lens_host=lens, n_tok_rows=n_tok_rows,
                full=full, side=side, tok_cnt=max(n_img, 1), shift_labels=shift.reshape(-1))
    tables = dict(kind=fk, row=fr, lens=lens, img_dst=img_dst, tok_src=tok_src.reshape(-1), embed_idx=embed_idx)
    # rows that carry a next-token label: the lm_head GEMMs and the cross-entropy only run on those (image / task-token / prompt /
    # pad rows have label IGNORE_INDEX: zero loss and zero d_logits in the reference's CrossEntropyLoss, ola_llama.py:127-136)
    ce_rows = np.flatnonzero(plan["shift_labels"] != IGNORE_INDEX).astype(np.int32)
    ce_inv = np.full(M, -1, np.int32)
    ce_inv[ce_rows] = np.arange(ce_rows.size, dtype=np.int32)
    plan["ce_labels"] = plan["shift_labels"][ce_rows]
    tables["ce_rows"], tables["ce_inv"] = ce_rows, ce_inv
    tables["ce_kind"], tables["ce_inv_kind"] = np.zeros(ce_rows.size, np.int32), np.where(ce_inv >= 0, 0, -1).astype(np.int32)
    # ... and the rows that carry none: when the caller wants the reference's `logits` of every row (ola_llama.py:121-122) they only need
    # the lm_head forward (no cross-entropy, no d_hidden GEMM)
    tables["un_rows"] = np.flatnonzero(plan["shift_labels"] == IGNORE_INDEX).astype(np.int32)
    if side == "left" and not full:
        src = col - (S - lens)[:, None]                          # presented column c shows physical column c - (S - len)
        ok = src >= 0
        present = np.where(ok, np.arange(B, dtype=np.int32)[:, None] * S + src, -1).astype(np.int32)
        plan["labels"] = np.where(ok, np.take_along_axis(lab2, np.maximum(src, 0).astype(np.int64), 1), IGNORE_INDEX)
        plan["attention_mask"] = ok.copy()
        plan["position_ids"] = np.where(ok, src, 0).astype(np.int64)
        tables["present"] = present.reshape(-1)
        tables["present_kind"] = np.where(present.reshape(-1) >= 0, 0, -1).astype(np.int32)
    else:
        plan["labels"] = lab2
        plan["attention_mask"] = real.copy()
        plan["position_ids"] = np.where(real, col, 0).astype(np.int64)
    heads = head_tables(cfg, tasks, B, S)
    if heads and side == "left" and not full:
        raise NotImplementedError(
            "distillation heads with a RAGGED left-padded batch: the reference slices head inputs by absolute position "
            "(base_ola_vlm.py:414-427), which only lines up with right padding (its training scripts' setting)")
    for task, h in heads.items():
        tables["rows:" + task] = h["rows_host"]
        tables["xin_kind:" + task], tables["xin_row:" + task] = h["xin_kind"], h["xin_row"]
        if h.get("mean_idx") is not None:
            tables["mean_idx:" + task] = h["mean_idx"]
        if "lat_bwd" in h:
            tables["lat_bwd:" + task] = h["lat_bwd"]
        if "lat_rep" in h:
            tables["lat_rep:" + task] = h["lat_rep"]
    for l, inv in inverse_tables(tasks, heads, M).items():
        tables[f"inv:{l}"] = inv.reshape(-1)
    plan["tables"] = tables
    plan["heads"] = heads
    return plan
