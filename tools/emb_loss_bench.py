"""HBM throughput of the distillation-loss reduction (vp_emb_loss_*) at the config-2 sizes (dev tool; gpurun).
Two clocks per shape: `loop` = python issuing back-to-back calls (includes the host's ~10-30 us of allocation + ctypes per call when the GPU
is faster than that), `graph` = the same calls captured once into a HIP graph and replayed (device time per launch incl. the launch gap)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops

N = 20
ONLY = os.environ.get("EL_ONLY", "")


def t_loop(fn, n=N):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def t_graph(fn, n=N, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def line(name, world, uf, ub, gf, gb, by_f, by_b):
    print(f"{name:5s} world={world}: fwd loop {uf:6.1f} us graph {gf:6.1f} us = {by_f / gf / 1e3:7.1f} GB/s ({by_f / gf / 8e6:.3f}) | "
          f"bwd loop {ub:6.1f} us graph {gb:6.1f} us = {by_b / gb / 1e3:7.1f} GB/s ({by_b / gb / 8e6:.3f})", flush=True)


B = int(os.environ.get("EL_B", "8"))
shapes = (("gen", 1024), ("depth", 576 * 1024), ("seg", 1536 * 576))
for world in (1, 8):
    Bw = B * world
    data = []
    for name, D in shapes:
        pred = torch.randn(B, D, device="cuda", dtype=torch.bfloat16)
        tgt = torch.randn(Bw, D, device="cuda", dtype=torch.bfloat16)
        mask = torch.ones(B, device="cuda")
        scale = torch.full((1,), 2.0, device="cuda")
        _, coef = ops.emb_loss_fwd(pred, tgt, mask, scale, 0.3, rank=0)
        data.append((pred, tgt, mask, scale, coef))
        f = lambda: ops.emb_loss_fwd(pred, tgt, mask, scale, 0.3, rank=0)
        b = lambda: ops.emb_loss_bwd(pred, tgt, coef, 0.5, rank=0)
        line(name, world, t_loop(f), t_loop(b), t_graph(f), t_graph(b), 2.0 * D * (B + Bw), 2.0 * D * (2 * B + Bw))
    preds, tgts, masks, scales, coefs = map(list, zip(*data))
    f = lambda: ops.emb_loss_fwd_multi(preds, tgts, masks, scales, [0.3] * 3)
    b = lambda: ops.emb_loss_bwd_multi(preds, tgts, coefs, [0.5] * 3)
    line("all3", world, t_loop(f), t_loop(b), t_graph(f), t_graph(b), sum(2.0 * D * (B + Bw) for _, D in shapes), sum(2.0 * D * (2 * B + Bw) for _, D in shapes))
