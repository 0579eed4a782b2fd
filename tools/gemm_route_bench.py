"""Small / awkward GEMM shapes of the step (ViT tower, heads): auto routing vs forcing the 128-tile kernel (dev tool; run through gpurun)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
dev = "cuda"
shapes = [(4616, 1024, 4096), (4616, 4096, 1024), (4616, 3072, 1024), (4616, 1024, 1024), (4608, 4096, 4096), (4608, 1536, 1536),
          (4096, 4096, 4608), (20864, 4096, 4096), (1536, 4096, 20864), (384, 4096, 20864), (1024, 4096, 16320), (3040, 4096, 128256)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = {}
    for name, force in (("auto", 0), ("k128", 2), ("k256", 3), ("p8", 7)):
        try:
            for _ in range(5):
                ops.gemm(a, w, out=o, force_generic=force)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.gemm(a, w, out=o, force_generic=force)
            e1.record(); torch.cuda.synchronize()
            row[name] = e0.elapsed_time(e1) / 30 * 1e3
        except Exception as e:
            row[name] = float("nan")
    fl = 2.0 * M * N * K
    print((M, N, K), {k: f"{v:.1f}us/{fl / v / 1e6:.0f}TF" for k, v in row.items()}, flush=True)
