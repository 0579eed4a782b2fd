#!/bin/bash
# kernel-level durations of the vp_emb_loss_* kernels in tools/emb_loss_bench.py (run via gpurun)
root=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -- python $root/tools/emb_loss_bench.py > /tmp/pe.log 2>&1
grep -v amdgpu /tmp/pe.log | tail -6
f=$(find /tmp/pe -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python - "$f" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "emb_loss" in r["Kernel_Name"]]
n = len(per) // 6
names = ["gen w1", "gen w8", "depth w1", "depth w8", "seg w1", "seg w8"]
for c in range(6):
    d = collections.defaultdict(list)
    for k, us in per[c * n:(c + 1) * n]: d[k].append(us)
    print(names[c], {k: round(sorted(v)[len(v) // 2], 1) for k, v in d.items()})
PY
fi
