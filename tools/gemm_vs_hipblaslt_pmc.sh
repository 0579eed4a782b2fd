#!/bin/bash
# Do hipBLASLt's kernels win on CYCLES or on CLOCK?  GRBM_GUI_ACTIVE (GPU-active cycles) / kernel duration = the clock a kernel sustains under
# the package power cap; SQ counters give the wave-cycle anatomy.  Run via gpurun; summary lands in gpurun_out/pmc_<tag>/.
tag=${1:-r03_vs_hipblaslt}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_vs
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pmc_vs -- python $root/tools/gemm_vs_hipblaslt_pmc.py > $out/run.log 2>&1
python - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
kt = glob.glob("/tmp/pmc_vs/**/*kernel_trace.csv", recursive=True)
cc = glob.glob("/tmp/pmc_vs/**/*counter_collection.csv", recursive=True)
rows = {}
for r in csv.DictReader(open(kt[0])):
    rows[int(r["Dispatch_Id"])] = dict(name=r["Kernel_Name"], ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), c={})
for r in csv.DictReader(open(cc[0])):
    d = int(r["Dispatch_Id"])
    if d in rows:
        rows[d]["c"][r["Counter_Name"]] = rows[d]["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
groups, prev = [], None
for d in sorted(rows):
    r = rows[d]
    if not ("gemm_nt" in r["name"] or "Cijk" in r["name"]):
        continue
    if prev != r["name"]:
        groups.append((r["name"], []))
        prev = r["name"]
    groups[-1][1].append(r)
CTR = ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU")
with open(out + "/summary.txt", "w") as fh:
    for name, rs in groups:
        rs = rs[len(rs) // 4:]                                  # drop the ramp-up quarter of each run of back-to-back launches
        ns = sum(r["ns"] for r in rs) / len(rs)
        line = f"{name[:48]:48s} n={len(rs):3d} avg_us={ns / 1e3:8.1f}"
        for k in CTR:
            v = [r["c"][k] for r in rs if k in r["c"]]
            if v:
                m = sum(v) / len(v)
                if k == "GRBM_GUI_ACTIVE":
                    line += f" gui_cycles_per_xcd={m / 8:.0f} clock_MHz={m / 8 / ns * 1e3:.0f}"
                else:
                    line += f" {k}={m:.3e}"
        print(line); fh.write(line + "\n")
PY
