"""HBM throughput of the distillation-loss reduction (vp_emb_loss_*) at the config-2 sizes (dev tool; gpurun)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = int(os.environ.get("EL_B", "8"))
for name, D in (("gen", 1024), ("depth", 576 * 1024), ("seg", 1536 * 576)):
    for world in (1, 8):
        Bw = B * world
        pred = torch.randn(B, D, device="cuda", dtype=torch.bfloat16)
        tgt = torch.randn(Bw, D, device="cuda", dtype=torch.bfloat16)
        mask = torch.ones(B, device="cuda")
        scale = torch.full((), 2.0, device="cuda")
        loss3, coef = ops.emb_loss_fwd(pred, tgt, mask, scale, 0.3, rank=0)
        ms_f = t(lambda: ops.emb_loss_fwd(pred, tgt, mask, scale, 0.3, rank=0))
        ms_b = t(lambda: ops.emb_loss_bwd(pred, tgt, coef, 0.5, rank=0))
        by_f = 2.0 * D * (B + Bw)                       # bf16 pred + all gathered targets read once
        by_b = 2.0 * D * (2 * B + Bw)                   # + dpred written
        print(f"{name:5s} D={D:7d} world={world}: fwd {ms_f * 1e3:7.1f} us {by_f / ms_f / 1e6:7.1f} GB/s | bwd {ms_b * 1e3:7.1f} us {by_b / ms_b / 1e6:7.1f} GB/s")
