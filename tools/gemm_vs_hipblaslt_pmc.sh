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
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X") or r.get("Grid_Size"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    d = r["Dispatch_Id"]
    if d not in dur:
        continue
    name, ns, grid = dur[d]
    if not ("gemm_nt" in name or "Cijk" in name):
        continue
    key = (name[:70], r.get("Grid_Size", grid))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_ns_" + d] = [ns]
with open(out + "/summary.txt", "w") as fh:
    for key, c in agg.items():
        nss = [v[0] for k, v in c.items() if k.startswith("_ns_")]
        nss = sorted(nss)[len(nss) // 4:]                       # drop the ramp-up quarter
        ns = sum(nss) / len(nss)
        line = f"{key[0]} grid={key[1]} n={len(nss)} avg_us={ns / 1e3:.1f}"
        for k in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU"):
            if k in c:
                v = c[k]
                v = sorted(v)[len(v) // 4:]
                m = sum(v) / len(v)
                line += f" {k}={m:.0f}"
                if k == "GRBM_GUI_ACTIVE":
                    line += f" clock_MHz={m / ns * 1e3:.0f}"
        print(line); fh.write(line + "\n")
PY
