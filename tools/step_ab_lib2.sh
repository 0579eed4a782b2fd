# Dev tool (gpurun): same-box A/B of the current library against TWO variant libraries (V1, V2 relative to the repo root), interleaved, two rounds.
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probes ${WL:+--workload $WL} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for r in 1 2; do
  run current
  VP_LIB_PATH=$GRAFT_REPO_ROOT/$V1 run variant1
  VP_LIB_PATH=$GRAFT_REPO_ROOT/$V2 run variant2
done
