#!/bin/bash
# GPU idle time inside the timed steps: kernel-trace of a short bench run, busy time vs span of the last step (dev tool; gpurun)
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python - "$f" <<PY
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps are delimited by the adamw kernel
ad = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
a, b = ad[-2] + 1, ad[-1]
seg = rows[a:b + 1]
span = seg[-1][1] - seg[0][0]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))
print(f"last step: {len(seg)} kernels, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.2f} %)")
print("largest gaps (us):", [(round(g, 1), x, y) for g, x, y in gaps[-6:]])
tot = {}
for s_, e_, n in seg:
    k = n[:60]
    t = tot.setdefault(k, [0, 0]); t[0] += e_ - s_; t[1] += 1
print("per-kernel totals inside the step (ms, calls):")
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:90]:
    print(f"  {t / 1e6:8.3f} {c:5d}  {k}")
PY
fi
