#!/bin/bash
# CU-side counters (VMEM / LDS issue, FIFO-full stalls, LDS conflicts) of our GEMM (FORCE env) vs hipBLASLt.  Run via gpurun.
tag=${1:-r03_sq}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  d=/tmp/pmc_sq_$(echo $pass | md5sum | cut -c1-8)
  rm -rf $d
  REPS=12 timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python $root/tools/gemm_vs_hipblaslt_pmc.py > $out/run.log 2>&1
  python - "$d" "$pass" <<'PY'
import csv, sys, glob
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
for gi, (name, rs) in enumerate(groups):
    if gi % 4 >= 2:
        continue                                              # each shape runs ours / theirs twice: print the first pair
    line = f"{name[:28]:28s}"
    for k in names:
        v = [r["c"][k] for r in rs if k in r["c"]]
        if v:
            line += f" {k.replace('SQ_', '')}={sum(v) / len(v):.3e}"
    print(line)
PY
done 2>&1 | tee $out/summary.txt
