#!/bin/bash
# Round 6: validation of the store-data keep-alive (gemm.hip W4_KEEP2) WITHOUT the whole-register-file claim (variants/libvisper_noclob_keep.so), the
# configuration in which the one-wave-per-SIMD GEMM used to go wrong beside other kernels; then the step time with and without the claim.
root=$(pwd); V=$root/visper-lm_amd/variants
pat() { VP_LIB_PATH=$1 timeout 600 python tools/nan_pattern_r06.py 2>&1 | grep -v amdgpu.ids | grep "library:\|launches differ"; }
ift() { env "$@" timeout 600 python bench.py --workload ift --steps 8 --warmup 2 --no-probes --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('   ift bench: loss', d['config']['loss'], 'ms/step', d['ms_per_step'])
except Exception as e:
    print('   ift bench: NO LINE:', l[-120:])
"; }
pt() { env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip()); print('   pt bench: loss', d['config']['loss'], 'ms/step', d['ms_per_step'], 'gemm frac', d['roofline']['frac'], d['roofline']['family']['frac'])"; }
echo "== no claim, no keep-alive (control)"; pat $V/libvisper_noclob.so; ift VP_LIB_PATH=$V/libvisper_noclob.so
echo "== no claim, WITH the keep-alive"; for r in 1 2 3; do pat $V/libvisper_noclob_keep.so; done; for r in 1 2 3 4; do ift VP_LIB_PATH=$V/libvisper_noclob_keep.so; done
LAYERS=32 STEPS=4 pat $V/libvisper_noclob_keep.so
echo "== product library (claim + keep-alive)"; pat $root/visper-lm_amd/libvisper_hip.so; ift VP_DUMMY=1
echo "== PT step time: product (claim + keep-alive) vs no claim + keep-alive, interleaved"
for r in 1 2 3; do echo " product:"; pt VP_DUMMY=1; echo " no claim:"; pt VP_LIB_PATH=$V/libvisper_noclob_keep.so; done
echo "== IFT step time"; for r in 1 2; do echo " product:"; ift VP_DUMMY=1; echo " no claim:"; ift VP_LIB_PATH=$V/libvisper_noclob_keep.so; done
