# Dev tool (gpurun): same-box A/B: current library vs the variant with the first-half-of-round-5 dQ kernel (one block per item, no prefetch), vs round 4's dQ.
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for r in 1 2; do
  VP_ATTN_BWD64=1 run current_mode1
  VP_LIB_PATH=$GRAFT_REPO_ROOT/visper-lm_amd/variants/libvisper_dq_r05a.so VP_ATTN_BWD64=1 run r05a_mode1
  VP_ATTN_BWD64=2 run current_mode2
  VP_LIB_PATH=$GRAFT_REPO_ROOT/visper-lm_amd/variants/libvisper_dq_r05a.so VP_ATTN_BWD64=0 run r05a_mode0
done
