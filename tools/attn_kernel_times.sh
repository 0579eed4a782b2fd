#!/bin/bash
# average kernel durations of tools/attn_bench.py under rocprofv3 (dev tool; run via gpurun)
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pa
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python $root/tools/attn_bench.py > /dev/null 2>&1
f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attn_' in r['Name']: print(r['Name'][:44], round(float(r['AverageNs'])/1e3,1), 'us')
"; fi
