"""Where a tile's time goes in the D = 128 forward: shader-cycle sums per phase of waves 0 and 7 of the heaviest block (dev tool; gpurun).
The stamps force the MFMA results / LDS traffic of each phase to complete, so phases that normally overlap are serialised: read it as an
upper bound per phase and for the barrier-wait share."""
import os, sys, ctypes as C
os.environ["VP_ATTN_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
os.environ.setdefault("VP_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visper-lm_amd", "libvisper_hip_debug.so"))   # vp_debug_* live in the -DVP_DEBUG build
from visper_lm_amd import ops, _lib
B, Hq, Hkv, S, D = 8, 32, 8, 2048, 128
qkv = torch.randn(B, S, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
for _ in range(5):
    ops.attn_fwd(q, k, v, True)
torch.cuda.synchronize()
buf = (C.c_long * 16)(); _lib.call("vp_debug_attn_stamps", buf)
a = np.array(buf[:], dtype=np.int64).reshape(2, 8)
names = ["issue loads", "QK^T (16 LDS reads + 16 MFMA)", "mask + softmax", "PV (32 tr reads + 16 MFMA)", "wait loads + LDS stores", "barrier"]
for w, wn in ((0, "wave 0"), (1, "wave 7")):
    n = max(int(a[w][7]), 1)
    print(wn, f"{n} tiles, cycles per tile:", {names[i]: round(float(a[w][i]) / n) for i in range(6)}, "total", round(float(a[w][:6].sum()) / n))
