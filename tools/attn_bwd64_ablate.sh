#!/bin/bash
# Dev tool: builds ablation variants of the one-wave-per-SIMD attention backward (attention_bwd64.h, -DB64_ABL=<mask>: 1 no softmax arithmetic,
# 2 no LDS fragment reads, 4 no LDS-DMA / barriers; results are wrong, timing only) into visper-lm_amd/variants/ and, with `run`, times each on the GPU.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/visper-lm_amd/csrc
if [ "$1" = "build" ]; then
  mkdir -p $root/visper-lm_amd/variants $cs/build_abl
  for m in ${2:-1 2 4 7}; do
    /opt/rocm/bin/hipcc -DVP_DEBUG -DB64_ABL=$m --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -mllvm -amdgpu-spill-vgpr-to-agpr=0 \
      -c $cs/attention.hip -o $cs/build_abl/attention_$m.o &
  done
  wait
  for m in ${2:-1 2 4 7}; do
    objs=$(ls $cs/build/*.o | grep -v attention.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/visper-lm_amd/variants/libvisper_abl$m.so $objs $cs/build_abl/attention_$m.o -ldl
  done
  ls -la $root/visper-lm_amd/variants/
else
  cd /tmp && export TMPDIR=/tmp
  for m in 0 ${2:-1 2 4 7}; do
    if [ $m = 0 ]; then unset VP_LIB_PATH; else export VP_LIB_PATH=$root/visper-lm_amd/variants/libvisper_abl$m.so; fi
    rm -rf /tmp/ab$m
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$m -- python $root/tools/attn_bwd_time.py > /tmp/ab$m.log 2>&1
    f=$(find /tmp/ab$m -name "*kernel_stats.csv" | head -1)
    echo "ABL=$m $(grep -v amdgpu.ids /tmp/ab$m.log | grep 'ms (dQ' | tail -1)"
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if '64w' in r['Name'] or 'attn_bwd_d' in r['Name']: print('   ', r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
  done
fi
