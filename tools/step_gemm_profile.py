"""Per-shape GEMM time inside one full train step (dev tool; run through gpurun)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from visper_lm_amd import ops
from visper_lm_amd import config as C
from visper_lm_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "llama3_8b"          # llama3_8b | convnext | phi3
cfg = {"llama3_8b": C.llama3_8b, "convnext": C.llama3_8b_convnext, "phi3": C.phi3_mini}[wl]()
eng = Engine(cfg); eng.init_random(0)
B, T = (4, 3497) if wl == "phi3" else (8, 1449)
batch = bench.make_batch(cfg, B, T, 0, torch.device("cuda"))
for _ in range(2):
    eng.train_step(batch); eng.optimizer_step(1e-3)
torch.cuda.synchronize()
ops.GEMM_PROF = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.train_step(batch); eng.optimizer_step(1e-3); e1.record(); torch.cuda.synchronize()
prof, ops.GEMM_PROF = ops.GEMM_PROF, None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for a, b, fl, shp, _kind in prof:
    r = agg[(shp, _kind)]; r[0] += 1; r[1] += a.elapsed_time(b); r[2] += fl
tot = sum(r[1] for r in agg.values())
print(f"step {e0.elapsed_time(e1):.1f} ms, GEMM {tot:.1f} ms in {len(prof)} launches")
for shp, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"{str(shp):28s} n={n:4d} {ms:8.2f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9:7.0f} TF/s")
