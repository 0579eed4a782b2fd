#!/bin/bash
# dev tool (gpurun): per-kernel average duration of tools/attn_bwd_time.py under rocprofv3 for several builds of the library (timing ablations)
root=$(pwd); cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  rm -rf /tmp/pa
  VP_LIB_PATH=$root/$l timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python $root/tools/attn_bwd_time.py > /dev/null 2>&1
  f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
  echo "== $l"
  if [ -n "$f" ]; then python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'attn_' in r['Name']: print('  ', r['Name'][:40], round(float(r['AverageNs'])/1e3,1), 'us')
"; fi
done
