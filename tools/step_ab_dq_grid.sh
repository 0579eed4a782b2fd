# Dev tool (gpurun): same-box A/B of the dQ kernel's grid (persistent = one block per CU vs one block per item): kernel rows from rocprofv3, then the step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 0 256; do
  for lib in "" $R/visper-lm_amd/variants/libvisper_abl8.so; do
    rm -rf /tmp/pg; if [ -n "$lib" ]; then export VP_LIB_PATH=$lib; else unset VP_LIB_PATH; fi
    VP_ATTN_DQ_GRID=$g timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python $R/tools/attn_bwd_time.py > /tmp/pg.log 2>&1
    f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1)
    python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if '64w' in r['Name']: print('grid=$g abl8=${lib:+1}', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
  done
done
unset VP_LIB_PATH
cd $R
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
VP_ATTN_DQ_GRID=256 run grid256_a
VP_ATTN_DQ_GRID=0 run grid0_a
VP_ATTN_DQ_GRID=256 run grid256_b
VP_ATTN_DQ_GRID=0 run grid0_b
