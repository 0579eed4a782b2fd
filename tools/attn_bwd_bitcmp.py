"""Dev tool (gpurun): attention backward of two libraries (VP_LIB_PATH) on the cases of attn_bwd64_check.py, compared bitwise.
    python tools/attn_bwd_bitcmp.py run <tag> ; python tools/attn_bwd_bitcmp.py cmp <tagA> <tagB>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import attn_bwd64_check as C

if sys.argv[1] == "run":
    from visper_lm_amd import ops
    out = {}
    for case in C.CASES:
        name, B, Hq, Hkv, Sq, Skv, D, causal, window, kvlen, rope = case
        q, k, v, do, kv = C.make(case)
        o, lse = ops.attn_fwd(q, k, v, causal, window=window, kv_len=kv)
        kw = dict(causal=causal, window=window, kv_len=kv)
        if rope:
            kw["rope"] = ops.rope_tables(max(Sq, Skv), D, 10000.0, q.device)
        dq, dk, dv = ops.attn_bwd(q, k, v, o, lse, do, **kw)
        torch.cuda.synchronize()
        out[name] = (dq.cpu(), dk.cpu(), dv.cpu())
    torch.save(out, f"/tmp/attn_bitcmp_{sys.argv[2]}.pt")
else:
    a, b = torch.load(f"/tmp/attn_bitcmp_{sys.argv[2]}.pt"), torch.load(f"/tmp/attn_bitcmp_{sys.argv[3]}.pt")
    for name in a:
        res = []
        for nm, x, y in zip(("dq", "dk", "dv"), a[name], b[name]):
            neq = (x.view(torch.int16) != y.view(torch.int16))
            res.append(f"{nm} {'identical' if not bool(neq.any()) else f'{int(neq.sum())} of {neq.numel()} differ, first at {tuple(neq.nonzero()[0].tolist())}'}")
        print(name, "|", " | ".join(res))
