"""Dev tool: same-box A/B of two GEMM configurations (VP_GEMM_DBG bit sets / force codes), alternating, 40 back-to-back launches per measurement.
usage: gemm_ab.py <forceA>:<dbgA> <forceB>:<dbgB> [MxNxK ...]   (+res: residual epilogue with --res)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops, _lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
res = "--res" in sys.argv
cfgs = [tuple(int(x, 0) for x in a.split(":")) for a in args[:2]]
shapes = [tuple(int(x) for x in a.split("x")) for a in args[2:]] or [(16384, 4096, 4096), (16384, 6144, 4096), (16384, 28672, 4096), (16384, 4096, 14336)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda", dtype=torch.bfloat16) if res else None
    t = {c: [] for c in cfgs}
    for rnd in range(3):
        for c in cfgs:
            _lib.call("vp_debug_gemm_flags", c[1])
            for _ in range(8):
                ops.gemm(a, w, out=out, residual=r, force_generic=c[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                ops.gemm(a, w, out=out, residual=r, force_generic=c[0])
            e1.record(); torch.cuda.synchronize()
            t[c].append(e0.elapsed_time(e1) / 40 * 1e3)
    _lib.call("vp_debug_gemm_flags", 0)
    print(M, N, K, "res" if res else "", {f"{c[0]}:{hex(c[1])}": [round(x, 1) for x in v] for c, v in t.items()}, flush=True)
