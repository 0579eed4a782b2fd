#!/bin/bash
# clocks / power reported by rocm-smi while the 8-phase GEMM runs back to back (dev tool; run via gpurun)
root=$(pwd)
python - <<PY &
import sys, time; sys.path.insert(0, "$root")
import torch
from visper_lm_amd import ops
a = torch.randn(16384, 4096, device="cuda", dtype=torch.bfloat16); w = torch.randn(28672, 4096, device="cuda", dtype=torch.bfloat16) * 0.05
out = torch.empty(16384, 28672, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(20): ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
PY
pid=$!
sleep 6
rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Temp|Max" | head -12
sleep 4
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -4
wait $pid
