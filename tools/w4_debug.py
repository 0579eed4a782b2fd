import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 128), (256, 256, 256), (512, 512, 128), (256, 768, 384)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
    ref = ops.gemm(a, w, force_generic=7).float()
    out = ops.gemm(a, w, force_generic=8).float()
    bad = (out - ref).abs() > 1e-2 * ref.abs().max()
    print(M, N, K, "bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  bad rows", rows[:8].tolist(), "...", rows[-4:].tolist(), "n", rows.numel(), " bad cols", cols[:8].tolist(), "...", cols[-4:].tolist(), "n", cols.numel())
        # which k-range is wrong? compare with partial products
        for k0 in range(0, K, 64):
            part = (a[:, k0:k0+64].float() @ w[:, k0:k0+64].float().t())
            r, c = int(rows[0]), int(cols[0])
            print("   k-tile", k0 // 64, "partial at first bad", float(part[r, c]))
        r, c = int(rows[0]), int(cols[0]); print("   out", float(out[r, c]), "ref", float(ref[r, c]))
