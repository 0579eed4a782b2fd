#!/bin/bash
# dev tool (gpurun): rebuilds emb_loss.hip with EL_INFLIGHT variants on the GPU box and sweeps the block-count knobs
cd "$(dirname "$0")/.."
for inf in ${EL_INFLIGHTS:-24}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wall -Wno-unused-function -DEL_INFLIGHT=$inf -DEL_G1=${EL_G1:-32} \
    -c visper-lm_amd/csrc/emb_loss.hip -o visper-lm_amd/csrc/build/emb_loss.o && make -s -C visper-lm_amd/csrc || exit 1
  for cap in ${EL_CAPS:-256}; do for spw in ${EL_SPWS:-2}; do
    echo "== inflight=$inf cap=$cap spw=$spw"
    VP_EL_NBLK=$cap VP_EL_SPW=$spw python tools/emb_loss_bench.py 2>&1 | grep -E "world"
  done; done
done
