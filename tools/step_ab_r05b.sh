# Dev tool (gpurun): same-box A/B of the attention-backward choices inside the configs[1] step (two rounds, interleaved).
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for r in 1 2; do
  VP_ATTN_BWD64=1 VP_ATTN_DQ_GRID=256 run mode1_grid256
  VP_ATTN_BWD64=1 VP_ATTN_DQ_GRID=0 run mode1_grid0
  VP_ATTN_BWD64=2 run mode2_r4dq_newdkdv
  VP_ATTN_BWD64=0 run mode0_r4_pair
done
