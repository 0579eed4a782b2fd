"""Dev tool (gpurun): what the fused SwiGLU epilogues cost — the same GEMM shape with the plain epilogue vs the fused forward / backward one,
40 back-to-back launches each, alternating, three rounds (us per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops, _lib
def t(fn, n=40):
    for _ in range(6): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
for (M, F, H) in [(16384, 14336, 4096), (16384, 8192, 3072)]:
    x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    wgu = torch.randn(2 * F, H, device="cuda", dtype=torch.bfloat16) * 0.02
    dy = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    wdT = torch.randn(F, H, device="cuda", dtype=torch.bfloat16) * 0.02
    gu, act = ops.gemm_swiglu_fwd(x, wgu)
    o1 = torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16); o2 = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
    for r in range(3):
      for fl in [int(x, 0) for x in os.environ.get("DBG_FLAGS", "0").split(",")]:
        _lib.call("vp_debug_gemm_flags", fl)
        print(hex(fl), (M, F, H), "fwd plain", t(lambda: ops.gemm(x, wgu, out=o1)), "fused", t(lambda: ops.gemm_swiglu_fwd(x, wgu)),
              "| bwd plain", t(lambda: ops.gemm(dy, wdT, out=o2)), "fused", t(lambda: ops.gemm_swiglu_bwd(dy, wdT, gu)), flush=True)
