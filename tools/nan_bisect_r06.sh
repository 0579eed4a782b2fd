#!/bin/bash
# (Historical: the variant libraries below were built from experiment knobs in gemm.hip — VP_W4_CLOBBER_MASK=2 / 5 (genonly / notgen), VP_W4_VMCNT0,
# VP_W4_M0NOP, VP_W4_VCCNOP — of which only VP_W4_CLOBBER_MASK and VP_W4_STORE_KEEP remain in the source; its output is profiles/r06_nan_root_cause.txt.)
# Round 6: bisecting the NaNs of the one-wave-per-SIMD GEMM without its whole-register-file claim (repro: the IFT bench line with
# variants/libvisper_noclob.so, tools/nan_repro_r06.sh).  Each line = one IFT bench run (8 timed steps) under one library variant / environment.
root=$(pwd)
V=$root/visper-lm_amd/variants
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --workload ift --steps 8 --warmup 2 --no-probes --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('[$tag] loss', d['config']['loss'], 'ms/step', d['ms_per_step'])
except Exception as e:
    print('[$tag] NO LINE:', l[-160:])
"
}
run "noclob (control: NaN expected)" VP_LIB_PATH=$V/libvisper_noclob.so
run "genonly: only the general instantiation claims the file" VP_LIB_PATH=$V/libvisper_genonly.so
run "notgen: lean + fold claim, general does not" VP_LIB_PATH=$V/libvisper_notgen.so
run "noclob + vmcnt(0) at the K-tile hand-over" VP_LIB_PATH=$V/libvisper_noclob_vm0.so
run "noclob + s_nop 3 behind every m0 write" VP_LIB_PATH=$V/libvisper_noclob_m0nop.so
run "noclob, tower on the main stream" VP_LIB_PATH=$V/libvisper_noclob.so VP_TOWER_STREAM=0
run "noclob, tower starts after the previous decoder FORWARD" VP_LIB_PATH=$V/libvisper_noclob.so VP_TOWER_AFTER=fwd
run "noclob, tower starts after the previous decoder BACKWARD" VP_LIB_PATH=$V/libvisper_noclob.so VP_TOWER_AFTER=bwd
run "noclob, general GEMMs on the 8-phase kernel (VP_GEMM_W4G=0)" VP_LIB_PATH=$V/libvisper_noclob.so VP_GEMM_W4G=0
run "noclob + s_nop 4 behind both VCC writes of w4_rows8_swap" VP_LIB_PATH=$V/libvisper_noclob_vccnop.so
run "noclob + s_nop 4 behind the VCC writes, again" VP_LIB_PATH=$V/libvisper_noclob_vccnop.so
run "noclob again (control)" VP_LIB_PATH=$V/libvisper_noclob.so
