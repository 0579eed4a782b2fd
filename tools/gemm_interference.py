"""Persistent GEMM next to a stand-in collective (dev tool; run through gpurun): a side stream holds `BLOCKS` CUs with 64 KB of LDS each
(so no 8-phase GEMM block fits beside them) for ~8 ms while the main stream runs decoder-shaped GEMMs; static vs dynamic tile assignment."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops

M, N, K = 16384, 4096, 4096
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
side = torch.cuda.Stream()


def run(blocks, n=12):
    for _ in range(3):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    if blocks:
        with torch.cuda.stream(side):
            ops._lib.call("vp_debug_occupy", blocks, 40_000_000, side.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dyn in (0, 1):
    ops.set_dynamic(bool(dyn))
    base = run(0)
    for blocks in (8, 16, 32):
        ms = run(blocks)
        print(f"dynamic={dyn} occupied CUs={blocks:3d}: {ms:.3f} ms per GEMM ({ms / base:.2f}x of the undisturbed {base:.3f} ms)", flush=True)
ops.set_dynamic(False)
