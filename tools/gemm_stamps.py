import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
os.environ["VP_GEMM_DBG"] = str(0x10000 + int(sys.argv[1]) if len(sys.argv) > 1 else 0x10000)
import torch, numpy as np
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops, _lib
lib = _lib.load()
SH = [tuple(int(x) for x in a.split("x")) for a in os.environ.get("STAMP_SHAPES", "16384x28672x4096").split(",")]
for (M, N, K) in SH:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out=out, force_generic=int(os.environ.get("STAMP_FORCE", "7")))
    torch.cuda.synchronize()
    buf = (C.c_long * 2048)()
    lib.vp_debug_stamps.argtypes = [C.c_void_p]
    rc = lib.vp_debug_stamps(buf)
    st = np.array(buf[:], dtype=np.int64).reshape(256, 8)[:min(256, (M // 256) * (N // 256))]      # only the blocks this launch ran
    t0 = st[:, 0].min()
    d = (st[:, :8] - t0) / 100.0   # us
    print(M, N, K, "rc", rc)
    print(" start  us: min %.1f max %.1f" % (d[:, 0].min(), d[:, 0].max()))
    print(" loop0 start: mean %.1f" % d[:, 1].mean())
    print(" first-tile K loop: mean %.1f min %.1f max %.1f" % ((d[:, 2] - d[:, 1]).mean(), (d[:, 2] - d[:, 1]).min(), (d[:, 2] - d[:, 1]).max()))
    print(" epilogue issue: mean %.2f min %.2f max %.2f" % ((d[:, 3] - d[:, 2]).mean(), (d[:, 3] - d[:, 2]).min(), (d[:, 3] - d[:, 2]).max()))
    print(" store drain: mean %.2f min %.2f max %.2f" % ((d[:, 4] - d[:, 3]).mean(), (d[:, 4] - d[:, 3]).min(), (d[:, 4] - d[:, 3]).max()))
    ep = d[:, 3] - d[:, 2]
    print(" epilogue by xcd:", [round(float(ep[x::8].mean()), 1) for x in range(8)])
    cyc = (st[:, 7] - st[:, 6]).astype(float); us = (st[:, 2] - st[:, 1]) / 100.0
    print(" shader clock during first K loop: %.0f MHz (cycles %.0f / %.1f us)" % ((cyc / us).mean(), cyc.mean(), us.mean()))
    print(" shader clock by xcd (MHz):", [int((cyc[x::8] / us[x::8]).mean()) for x in range(8)], " first K loop us by xcd:", [round(float(us[x::8].mean()), 1) for x in range(8)])
    e = np.sort(d[:, 5])
    print(" block end times (us) pct 0/10/50/90/100: %.1f %.1f %.1f %.1f %.1f ; by xcd mean:" % (e[0], e[len(e) // 10], e[len(e) // 2], e[len(e) * 9 // 10], e[-1]), [round(float(d[x::8, 5].mean()), 1) for x in range(8)])
    print(" end: mean %.1f max %.1f ; loopend spread %.1f..%.1f" % (d[:, 5].mean(), d[:, 5].max(), d[:, 2].min(), d[:, 2].max()))
