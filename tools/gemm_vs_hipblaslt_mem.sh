#!/bin/bash
# Memory-side counters of our GEMM (FORCE env: 0 default, 8 four-wave) vs hipBLASLt on the decoder shapes: fabric bytes and L2 hit rate.
tag=${1:-r03_mem}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  d=/tmp/pmc_mem_$(echo $pass | tr ' ' '_')
  rm -rf $d
  REPS=12 timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python $root/tools/gemm_vs_hipblaslt_pmc.py > $out/run_$(echo $pass | tr ' ' '_').log 2>&1
  python - "$d" "$pass" <<'PY'
import csv, sys, glob, collections
d, names = sys.argv[1], sys.argv[2].split()
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True); cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not kt or not cc:
    print("no output for", names); sys.exit(0)
rows = {}
for r in csv.DictReader(open(kt[0])):
    rows[int(r["Dispatch_Id"])] = dict(name=r["Kernel_Name"], c={})
for r in csv.DictReader(open(cc[0])):
    i = int(r["Dispatch_Id"])
    if i in rows:
        rows[i]["c"][r["Counter_Name"]] = rows[i]["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
groups, prev = [], None
for i in sorted(rows):
    r = rows[i]
    if not ("gemm_nt" in r["name"] or "Cijk" in r["name"]):
        continue
    if prev != r["name"]:
        groups.append((r["name"], [])); prev = r["name"]
    groups[-1][1].append(r)
for name, rs in groups:
    line = f"{name[:40]:40s} n={len(rs):3d}"
    for k in names:
        v = [r["c"][k] for r in rs if k in r["c"]]
        if v:
            line += f" {k}={sum(v) / len(v):.4e}"
    print(line)
PY
done 2>&1 | tee $out/summary.txt
