"""Dev tool (gpurun): where the forward attention's time goes at block seams — causal vs non-causal at the decoder shape, and a short-S case
(blocks of one to four tiles: mostly prologue + epilogue).  us per launch, block-tiles (256 queries x 64 keys) and ns per block-tile and CU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
def run(B, Hq, Hkv, S, D, causal):
    qkv = torch.randn(B, S, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
    q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
    for _ in range(10): ops.attn_fwd(q, k, v, causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.attn_fwd(q, k, v, causal)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nqb = S // 256
    tiles = B * Hq * (sum((qb + 1) * 4 for qb in range(nqb)) if causal else nqb * (S // 64))
    blocks = B * Hq * nqb
    print(f"B={B} H={Hq} S={S} causal={causal}: {us:8.1f} us, {blocks} blocks, {tiles} block-tiles, {us * 256 / tiles * 1e3:7.0f} ns per block-tile and CU, "
          f"{us * 256 / blocks:6.2f} us per block and CU", flush=True)
run(8, 32, 8, 2048, 128, True)
run(8, 32, 8, 2048, 128, False)
run(8, 32, 8, 4096, 128, True)
run(64, 32, 8, 256, 128, True)      # one 4-tile block per (b, h): 2048 blocks
run(64, 32, 8, 256, 128, False)
run(16, 32, 8, 1024, 128, True)
