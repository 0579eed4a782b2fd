#!/bin/bash
# MFMA-utilisation / wait counters of the GEMM and attention kernels and the HBM rate of the loss reduction (run via gpurun).
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
BENCH_ONLY=${PMC_GEMM_VARIANT:-k256w4} timeout 600 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_gemm -- python $root/tools/gemm_bench.py 16384x4096x4096 16384x28672x4096 16384x4096x14336 > $out/gemm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_attn -- python $root/tools/attn_bench.py > $out/attn.log 2>&1
for d in gemm attn; do
  f=$(find /tmp/pmc_$d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then for c in $CTR; do python $root/tools/pmc_summarize.py "$f" $c | grep -E "^#|gemm_nt|attn_" ; done > $out/pmc_$d.txt; fi
done
python $root/tools/emb_loss_bench.py > $out/emb_loss.txt 2>&1
cat $out/pmc_gemm.txt $out/pmc_attn.txt $out/emb_loss.txt | grep -v amdgpu
