#!/bin/bash
# which hipBLASLt kernels torch.matmul picks for the train-step GEMM shapes (comparison baseline only; dev tool, run via gpurun)
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/hb
cat > /tmp/hb.py <<'PY'
import torch
for (M, N, K) in [(16384, 28672, 4096), (16384, 4096, 14336), (16384, 4096, 4096), (16384, 6144, 4096)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    for _ in range(3): torch.matmul(a, w.t())
    torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hb -- python /tmp/hb.py > /dev/null 2>&1
f=$(find /tmp/hb -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "Cijk" in r["Name"] or "gemm" in r["Name"].lower():
        print(round(float(r["AverageNs"]) / 1e3, 1), "us", r["Calls"], r["Name"])
PY
