run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
run base1
VP_ATTN_BWD64=2 run bwd64_2
VP_ATTN_BWD64=0 run bwd64_0
VP_GEMM_DYN=1 run gemm_dyn
VP_ATTN_ORDER=1 run attn_order1
run base2
VP_ATTN_BWD64=2 run bwd64_2b
