"""Dev aid (gpurun): dump the final statistics of vp_emb_loss_fwd (PT / TT / PP / SL in the workspace tail) against fp64 torch."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops, _lib
import ctypes as C


def plan(B, Bw, D):
    npb = 1 if B <= 16 else (2 if B <= 32 else 4)
    groups = (Bw + 15) // 16
    ng = 1 if groups <= 1 else (2 if groups <= 2 else (4 if groups <= 4 else 8))
    njc = (groups + ng - 1) // ng
    nsteps = (D + 31) // 32
    nblk = max(1, min(1024, (nsteps + 7) // 8))
    while njc * ((nblk + 31) // 32) > 1024:
        nblk //= 2
    ngrp = (nblk + 31) // 32
    ns = npb * 16 * ng * 16 + ng * 16 + 2 * npb * 16
    return npb, ng, njc, nblk, ngrp, ns


for (B, world, rank, D) in ((2, 4, 2, 2304), (8, 1, 0, 1024), (3, 1, 0, 40960), (8, 8, 3, 589824), (32, 8, 5, 2560)):
    Bw = B * world
    g = torch.Generator(device="cuda").manual_seed(1)
    pred = (torch.randn(B, D, device="cuda", generator=g) * 1.3).bfloat16()
    tgt = torch.randn(Bw, D, device="cuda", generator=g).bfloat16()
    mask = torch.ones(B, device="cuda")
    ls = torch.tensor([2.0], device="cuda")
    npb, ng, njc, nblk, ngrp, ns = plan(B, Bw, D)
    nws = _lib.raw("vp_emb_loss_workspace", B, Bw, D)
    for trial in range(3):
        ws = torch.full((nws,), float("nan"), device="cuda")
        coef = torch.empty(2 * B + B * Bw + 1, device="cuda")
        out3 = torch.empty(3, device="cuda")
        _lib.call("vp_emb_loss_fwd", B, Bw, D, rank, ops._p(pred), ops._p(tgt), ops._p(mask), ops._p(ls), 0.3, ops._p(out3), ops._p(coef), ops._p(ws), ops._loss_counters(), ops._stream())
        torch.cuda.synchronize()
        fin = ws[njc * nblk * ns + njc * ngrp * ns:]
        if fin.numel() < B * Bw + Bw + 2 * B:          # LDS-resident finalize: the sums never reach the workspace
            print(f"B={B} Bw={Bw} D={D} plan={plan(B, Bw, D)}: sums stay in LDS; out3 {out3.tolist()}")
            continue
        PT = fin[:B * Bw].view(B, Bw).double()
        TT = fin[B * Bw:B * Bw + Bw].double()
        PP = fin[B * Bw + Bw:B * Bw + Bw + B].double()
        SL = fin[B * Bw + Bw + B:B * Bw + Bw + 2 * B].double()
        p64, t64 = pred.double(), tgt.double()
        d = (p64 - t64[rank * B:(rank + 1) * B]).abs()
        rPT, rTT, rPP, rSL = p64 @ t64.t(), (t64 * t64).sum(1), (p64 * p64).sum(1), torch.where(d < 1, 0.5 * d * d, d - 0.5).sum(1)
        e = lambda a, b: float((a - b).abs().max() / b.abs().max())
        print(f"B={B} Bw={Bw} D={D} plan={plan(B, Bw, D)} trial {trial}: PT {e(PT, rPT):.2e} TT {e(TT, rTT):.2e} PP {e(PP, rPP):.2e} SL {e(SL, rSL):.2e} "
              f"out3 {out3.tolist()} dls {float(coef[-1]):.6f}")

# ---- in-kernel phase stamps (100 MHz wall clock) of the finishing block
stamps = torch.zeros(8, dtype=torch.int64, device="cuda")
_lib.call("vp_debug_emb_loss_stamps", ops._p(stamps))
for (B, world, D) in ((8, 1, 1024), (8, 1, 589824), (8, 8, 589824), (8, 1, 884736), (8, 8, 884736)):
    Bw = B * world
    pred = torch.randn(B, D, device="cuda").bfloat16(); tgt = torch.randn(Bw, D, device="cuda").bfloat16()
    mask = torch.ones(B, device="cuda"); ls = torch.tensor([2.0], device="cuda")
    for _ in range(3):
        ops.emb_loss_fwd(pred, tgt, mask, ls, 0.3)
    torch.cuda.synchronize()
    t = stamps.tolist()
    us = lambda a, b: (b - a) / 100.0
    print(f"B={B} Bw={Bw} D={D}: block0 start -> finisher start {us(t[0], t[1]):.1f} us | finisher: stream {us(t[1], t[2]):.1f}  lds+partial+ticket {us(t[2], t[3]):.1f} "
          f" tree {us(t[3], t[4]):.1f}  finalize {us(t[4], t[5]):.1f} | first start -> end {us(t[0], t[5]):.1f} us")
_lib.call("vp_debug_emb_loss_stamps", None)
