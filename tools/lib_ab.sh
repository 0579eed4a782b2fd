#!/bin/bash
# same-box A/B of two builds of the library (dev tool; run through gpurun): tools/lib_ab.sh <libA.so> <libB.so> <python script + args ...>
# alternates A B A B; the script must print its own timings
a=$1; b=$2; shift 2
for r in 1 2; do
  for l in $a $b; do echo "== $l"; VP_LIB_PATH=$PWD/$l python "$@" 2>&1 | grep -v amdgpu.ids; done
done
