#!/bin/bash
# Round profile (run through gpurun from the repo root): kernel-trace stats of the default bench, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) for the HBM traffic of the dominant kernel.  Outputs land in gpurun_out/prof_<tag>/.
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $root/bench.py --steps 3 --warmup 1 --no-probes --no-cpu-baseline > $out/bench_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probes > $out/bench_$c.log 2>&1
  f=$(find /tmp/prof_$c -name "*counter_collection.csv" | head -1); if [ -n "$f" ]; then python $root/tools/pmc_summarize.py "$f" $c > $out/pmc_$c.txt; fi
done
tail -1 $out/bench_stats.log | cut -c1-400
cat $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt 2>/dev/null | head -40
