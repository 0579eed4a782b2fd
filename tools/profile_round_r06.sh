#!/bin/bash
# Round-6 profile (run through gpurun from the repo root): (1) kernel trace of the default bench path (model API) with markers around the timed
# region -> per-kernel statistics of the TIMED STEPS ONLY (tools/kernel_stats_steps.py) next to rocprofv3's own --stats summary;
# (2) separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> <tag>_pmc_w4.json.  Outputs: gpurun_out/prof_<tag>/.
tag=${1:-r06}
steps=${2:-4}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $root/bench.py --steps $steps --warmup 2 --no-probes --no-cpu-baseline --no-extras --trace-markers > $out/bench_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $out/${tag}_bench_kernel_stats_whole_process.csv; fi
f=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1); if [ -n "$f" ]; then python $root/tools/kernel_stats_steps.py "$f" $steps $out/${tag}_bench_kernel_stats.csv > $out/kernel_stats_steps.log 2>&1; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probes --no-extras > $out/bench_$c.log 2>&1
  f=$(find /tmp/prof_$c -name "*counter_collection.csv" | head -1); if [ -n "$f" ]; then python $root/tools/pmc_summarize.py "$f" $c > $out/pmc_$c.txt; cp "$f" /tmp/cc_$c.csv; fi
done
python $root/tools/pmc_w4_json.py /tmp/cc_FETCH_SIZE.csv /tmp/cc_WRITE_SIZE.csv $tag > $out/${tag}_pmc_w4.json 2> $out/pmc_json.err
tail -1 $out/bench_stats.log | cut -c1-300
cat $out/kernel_stats_steps.log | head -20
head -12 $out/${tag}_pmc_w4.json; cat $out/pmc_json.err
