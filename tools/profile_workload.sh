#!/bin/bash
# kernel-trace stats of one bench workload (run through gpurun from the repo root): tools/profile_workload.sh <workload> [steps]
wl=${1:-convnext}; steps=${2:-2}
root=$(pwd); out=$root/gpurun_out/prof_$wl; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$wl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python $root/bench.py --workload $wl --steps $steps --warmup 1 --no-probes --no-cpu-baseline > $out/bench.log 2>&1
f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; fi
tail -1 $out/bench.log | cut -c1-300
