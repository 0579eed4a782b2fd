#!/bin/bash
# PMC counters of the attention kernels at the decoder shape (tools/attn_bench.py), two passes; run via gpurun.  $1 = tag
tag=${1:-r02}
root=$(pwd); out=$root/gpurun_out/pmc_attn_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"
i=0
for CTR in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pa$i
  timeout 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pa$i -- python $root/tools/attn_bench.py > $out/run$i.log 2>&1
  f=$(find /tmp/pa$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then for c in $CTR; do python $root/tools/pmc_summarize.py "$f" $c | grep -E "^#|attn_"; done >> $out/pmc.txt; fi
done
cat $out/pmc.txt
