"""Dev tool (gpurun): the one-wave-per-SIMD attention backward (VP_ATTN_BWD64=1) against round 4's kernels (=0) and against fp32 math.
    python tools/attn_bwd64_check.py run <tag>      # one process per kernel family (the switch is read once): dumps results under /tmp
    python tools/attn_bwd64_check.py cmp            # compares the two dumps + an fp32 reference on the small cases
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

CASES = [  # (name, B, Hq, Hkv, Sq, Skv, D, causal, window, kvlen, rope)
    ("small_causal", 2, 4, 2, 320, 320, 128, True, 0, None, False),
    ("ragged", 2, 4, 2, 300, 300, 128, True, 0, [300, 170], False),
    ("noncausal", 1, 4, 4, 288, 416, 128, False, 0, None, False),
    ("rope", 2, 8, 2, 512, 512, 128, True, 0, None, True),
    ("d96_window", 1, 4, 4, 700, 700, 96, True, 300, None, True),
    ("decoder", 8, 32, 8, 2048, 2048, 128, True, 0, None, True),
]


def make(case):
    name, B, Hq, Hkv, Sq, Skv, D, causal, window, kvlen, rope = case
    g = torch.Generator(device="cuda").manual_seed(sum(ord(ch) for ch in name))
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v, do = rn(B, Sq, Hq, D), rn(B, Skv, Hkv, D), rn(B, Skv, Hkv, D), rn(B, Sq, Hq, D)
    kv = None if kvlen is None else torch.tensor(kvlen, device="cuda", dtype=torch.int32)
    return q, k, v, do, kv


def run(tag):
    from visper_lm_amd import ops
    out = {}
    for case in CASES:
        name, B, Hq, Hkv, Sq, Skv, D, causal, window, kvlen, rope = case
        q, k, v, do, kv = make(case)
        o, lse = ops.attn_fwd(q, k, v, causal, window=window, kv_len=kv)
        kw = dict(causal=causal, window=window, kv_len=kv)
        if rope:
            cos_t, sin_t = ops.rope_tables(max(Sq, Skv), D, 10000.0, q.device)
            kw["rope"] = (cos_t, sin_t)
        dq, dk, dv = ops.attn_bwd(q, k, v, o, lse, do, **kw)
        torch.cuda.synchronize()
        ms = None
        if name == "decoder":
            for _ in range(3):
                ops.attn_bwd(q, k, v, o, lse, do, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.attn_bwd(q, k, v, o, lse, do, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
        print(tag, name, "finite", bool(torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all() and torch.isfinite(dv.float()).all()),
              "ms", ms, flush=True)
        out[name] = (dq.cpu(), dk.cpu(), dv.cpu(), o.cpu(), lse.cpu())
    torch.save(out, f"/tmp/attn_bwd64_{tag}.pt")


def ref32(case):
    name, B, Hq, Hkv, Sq, Skv, D, causal, window, kvlen, rope = case
    q, k, v, do, kv = make(case)
    qf, kf, vf = (t.float().cpu().requires_grad_(True) for t in (q, k, v))
    rep = Hq // Hkv
    kk, vv = kf.repeat_interleave(rep, 2), vf.repeat_interleave(rep, 2)
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kk) / D ** 0.5
    qi, ki = torch.arange(Sq)[:, None], torch.arange(Skv)[None, :]
    off = Skv - Sq
    m = torch.ones(Sq, Skv, dtype=torch.bool)
    if causal:
        m &= ki <= qi + off
    if window > 0:
        m &= ki > qi + off - window
    m = m[None, None].expand(B, 1, Sq, Skv).clone()
    if kvlen is not None:
        for b_, L in enumerate(kvlen):
            m[b_, :, :, L:] = False
    s = s.masked_fill(~m, float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    o = torch.einsum("bhqk,bkhd->bqhd", p, vv)
    o.backward(do.float().cpu())
    return qf.grad, kf.grad, vf.grad


def cmp():
    a, b = torch.load("/tmp/attn_bwd64_new.pt"), torch.load("/tmp/attn_bwd64_old.pt")
    rel = lambda x, y: float((x.float() - y.float()).abs().max() / y.float().abs().max().clamp_min(1e-30))
    for case in CASES:
        name, rope = case[0], case[-1]
        na, nb = a[name], b[name]
        line = f"{name}: new-vs-old max-rel dq {rel(na[0], nb[0]):.2e} dk {rel(na[1], nb[1]):.2e} dv {rel(na[2], nb[2]):.2e}"
        if not rope and name != "decoder":
            t0 = time.time()
            rq, rk, rv = ref32(case)
            line += f" | vs fp32: new dq {rel(na[0], rq):.2e} dk {rel(na[1], rk):.2e} dv {rel(na[2], rv):.2e}; old dq {rel(nb[0], rq):.2e} dk {rel(nb[1], rk):.2e} dv {rel(nb[2], rv):.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        cmp()
