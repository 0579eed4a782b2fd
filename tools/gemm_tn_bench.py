"""TN (transpose-free wgrad) GEMM micro-benchmark on the IFT weight-gradient shapes (dev tool; run through gpurun).
Compares vp_gemm_tn_bf16 with the transposes + NT GEMM it replaces; fp32 output."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops

# (out features M, in features N, tokens K)
SHAPES = [(4096, 4096, 16384), (6144, 4096, 16384), (28672, 4096, 16384), (4096, 14336, 16384), (4096, 1024, 4608),
          (128256, 4096, 2048), (512, 512, 4096)]


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
    dy = torch.randn(K, M, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ref = dy[:, :256].float().t() @ x.float()
    ops.gemm_tn(dy, x, out=out)
    err = float((out[:256] - ref).abs().max() / ref.abs().max())
    ms_tn = timeit(lambda: ops.gemm_tn(dy, x, out=out))

    def via_nt():
        ops.gemm(ops.transpose(dy), ops.transpose(x), out=out, out_f32=True)
    ms_nt = timeit(via_nt)
    dyt, xt = ops.transpose(dy), ops.transpose(x)
    ms_nt_only = timeit(lambda: ops.gemm(dyt, xt, out=out, out_f32=True))
    fl = 2.0 * M * N * K / 1e9
    print(f"{M}x{N}x{K}: " + json.dumps({"tn_tf": round(fl / ms_tn, 1), "err": round(err, 6), "transposes+nt_tf": round(fl / ms_nt, 1),
                                          "nt_only_tf": round(fl / ms_nt_only, 1), "ms_tn": round(ms_tn, 3), "ms_old": round(ms_nt, 3)}),
          flush=True)
