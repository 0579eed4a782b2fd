#!/bin/bash
# Dev tool (gpurun): shader clock the attention backward kernels sustain (GRBM_GUI_ACTIVE / kernel time) and their matrix-pipe utilisation.
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for m in 0 ${1:-}; do
  if [ $m = 0 ]; then unset VP_LIB_PATH; else export VP_LIB_PATH=$root/visper-lm_amd/variants/libvisper_abl$m.so; fi
  rm -rf /tmp/pc$m
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pc$m -- python $root/tools/attn_bwd_time.py > /tmp/pc$m.log 2>&1
  f=$(find /tmp/pc$m -name "*counter_collection.csv" | head -1)
  python3 - "$f" $m <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'attn_bwd' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append((float(r['Counter_Value']), float(r['End_Timestamp'])-float(r['Start_Timestamp'])))
for k,d in acc.items():
    ns=sum(t for _,t in d['GRBM_GUI_ACTIVE'][20:])/len(d['GRBM_GUI_ACTIVE'][20:])
    g=sum(x for x,_ in d['GRBM_GUI_ACTIVE'][20:])/len(d['GRBM_GUI_ACTIVE'][20:])
    av=lambda n: sum(x for x,_ in d[n][20:])/max(1,len(d[n][20:]))
    clk=g/ns
    print(f"ABL={sys.argv[2]} {k}: {ns/1e3:.1f} us, clock {clk:.2f} GHz, MFMA busy {av('SQ_VALU_MFMA_BUSY_CYCLES')/(g*1024):.3f} of SIMD-cycles, wave quad-cycles {av('SQ_WAVE_CYCLES'):.3g}: parked {av('SQ_WAIT_ANY')/av('SQ_WAVE_CYCLES'):.3f} stalled {av('SQ_WAIT_INST_ANY')/av('SQ_WAVE_CYCLES'):.3f} issuing {av('SQ_ACTIVE_INST_ANY')/av('SQ_WAVE_CYCLES'):.3f}")
PY
done
