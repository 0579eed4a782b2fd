#!/bin/bash
# Round 6, VERDICT r5 next-5: does the one-wave-per-SIMD GEMM WITHOUT its whole-register-file claim (variants/libvisper_noclob.so: gemm.hip built
# with -DVP_W4_NO_CLOBBER, 464 / 432 / 488 registers per wave) still go wrong beside other kernels?  Three repro configurations of round 4, each
# with the product library (control) and the variant.  Run through gpurun from the repo root.
root=$(pwd)
out=$root/gpurun_out/nan_repro
mkdir -p $out
run() {  # tag, env...
  tag=$1; shift
  for rep in 1 2 3; do
    env "$@" timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "w4_gemm_beside_small_kernels" 2>&1 | tail -1 | sed "s/^/[$tag rep $rep stress] /"
  done
  env "$@" timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -k "overlapped_steps_equal_synchronised or side_stream_schedule_is_bit_identical" 2>&1 | tail -1 | sed "s/^/[$tag overlapped-vs-synchronised] /"
  env "$@" timeout 900 python -m pytest tests/test_rccl_gpu.py -q -x -k "1000_rccl" 2>&1 | tail -1 | sed "s/^/[$tag rccl-1000] /"
  for rep in 1 2; do
    env "$@" timeout 600 python bench.py --workload ift --steps 8 --warmup 2 --no-probes --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('[$tag rep $rep ift bench] loss', d['config']['loss'], 'ms/step', d['ms_per_step'])
except Exception as e:
    print('[$tag rep $rep ift bench] NO LINE:', l[-200:])
"
  done
}
run product VP_DUMMY=1 2>&1 | tee $out/product.log
run noclob VP_LIB_PATH=$root/visper-lm_amd/variants/libvisper_noclob.so 2>&1 | tee $out/noclob.log
