// Dev probe: cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD when NCH independent accumulator chains alternate (NCH = 1, 2, 3, 4, 8), and
// with F filler VALU instructions per MFMA (0, 2, 4).  Answers: does a 2-chain phase (S^T of two q blocks) run at the 32-cycle issue rate?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o /tmp/mfma_chain_probe && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCH, int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(long* out, int iters) {
  asm volatile("" ::: "v255", "a255");
  for (int i = 0; i < 128; ++i) {}
  asm volatile("s_nop 0");
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      constexpr int dummy = 0;
      switch (u % NCH) {
        case 0: asm volatile("v_mfma_f32_32x32x16_bf16 v[64:79], v[200:203], v[204:207], v[64:79]"); break;
        case 1: asm volatile("v_mfma_f32_32x32x16_bf16 v[80:95], v[200:203], v[204:207], v[80:95]"); break;
        case 2: asm volatile("v_mfma_f32_32x32x16_bf16 v[96:111], v[200:203], v[204:207], v[96:111]"); break;
        case 3: asm volatile("v_mfma_f32_32x32x16_bf16 v[112:127], v[200:203], v[204:207], v[112:127]"); break;
        case 4: asm volatile("v_mfma_f32_32x32x16_bf16 v[128:143], v[200:203], v[204:207], v[128:143]"); break;
        case 5: asm volatile("v_mfma_f32_32x32x16_bf16 v[144:159], v[200:203], v[204:207], v[144:159]"); break;
        case 6: asm volatile("v_mfma_f32_32x32x16_bf16 v[160:175], v[200:203], v[204:207], v[160:175]"); break;
        default: asm volatile("v_mfma_f32_32x32x16_bf16 v[176:191], v[200:203], v[204:207], v[176:191]"); break;
      }
      if (F >= 2) { asm volatile("v_mul_f32 v210, v210, v211"); asm volatile("v_exp_f32 v212, v212"); }
      if (F >= 4) { asm volatile("v_mul_f32 v213, v213, v211"); asm volatile("v_exp_f32 v214, v214"); }
    }
  }
  long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NCH, int F>
void run(long* d) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NCH, F>), dim3(256), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NCH, F>), dim3(256), dim3(256), 0, 0, d, iters);
  long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("chains %d fillers/MFMA %d: %.1f clock64 ticks per MFMA\n", NCH, F, (double)h / (iters * 16.0));
}
int main() {
  long* d;
  hipMalloc(&d, 64);
  run<1, 0>(d); run<2, 0>(d); run<3, 0>(d); run<4, 0>(d); run<8, 0>(d);
  run<2, 2>(d); run<4, 2>(d); run<8, 2>(d); run<2, 4>(d); run<4, 4>(d); run<8, 4>(d);
  return 0;
}
