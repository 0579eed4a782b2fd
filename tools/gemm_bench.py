"""GEMM micro-benchmark on the real train-step shapes (dev tool; run through gpurun).
Compares the 128-tile and 256-tile kernels (and hipBLASLt via torch.matmul as a comparison baseline only),
random N(0,1) operands, within-process interleaved rounds (median)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops

SHAPES = [(16384, 6144, 4096), (16384, 4096, 4096), (16384, 28672, 4096), (16384, 4096, 14336),
          (16384, 14336, 4096), (16384, 4096, 28672), (16384, 4096, 6144), (2048, 128256, 4096), (2048, 4096, 128256),
          (4096, 4096, 4096), (8192, 8192, 8192)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = (a[:256].float() @ w.float().t())
    res = {}
    variants = (("k128", 2), ("k256", 3), ("k256p8", 7), ("k256w4", 8), ("k256p4", 13), ("auto", 0))
    only = os.environ.get("BENCH_ONLY")
    if only:
        variants = tuple(v for v in variants if v[0] in only.split(","))
    for name, fg in variants:
        ops.gemm(a, w, out=out, force_generic=fg)
        err = float((out[:256].float() - ref).abs().max() / ref.abs().max())
        ms = timeit(lambda: ops.gemm(a, w, out=out, force_generic=fg))
        res[name] = (round(2.0 * M * N * K / ms / 1e9, 1), round(err, 5))
    if not only:
        ms = timeit(lambda: torch.matmul(a, w.t(), out=out))
        res["hipblaslt"] = round(2.0 * M * N * K / ms / 1e9, 1)
    print(f"{M}x{N}x{K}: " + json.dumps(res), flush=True)
