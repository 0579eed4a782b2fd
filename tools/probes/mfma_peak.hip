// MFMA-only throughput probe (dev tool): 8 waves per block, 1 block per CU, 32 independent 16x16x32 bf16 accumulators
// per wave, operands in registers, no LDS / barriers.  Prints achieved TFLOP/s for 16x16x32 and 32x32x16.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// random bf16 in (-2, 2) (random sign, exponent 0x3e..0x3f, random mantissa) when seed0 != 0: data-dependent power is real (zeroed / constant
// operands clock ~19 % higher: MI355X_MICROARCH.md), so the probe runs on random bits like the GEMM does
__device__ inline short rnd_bf16(unsigned k, short keep) {
  if (keep == 0) return 0;
  k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
  return (short)(((k & 1u) << 15) | (0x3e80u + ((k >> 1) & 0xffu)) | ((k >> 9) & 0x7fu));
}
__global__ __launch_bounds__(512) void k16(float* out, int iters, bf16x8 a0, bf16x8 b0) {
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) { a[i] = a0; for (int e = 0; e < 8; ++e) a[i][e] = rnd_bf16(threadIdx.x * 131 + i * 17 + e, a0[e]); }
  for (int j = 0; j < 4; ++j) { b[j] = b0; for (int e = 0; e < 8; ++e) b[j][e] = rnd_bf16(threadIdx.x * 257 + j * 29 + e + 7, b0[e]); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i * 4 + j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k32(float* out, int iters, bf16x8 a0, bf16x8 b0) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) { a[i] = a0; for (int e = 0; e < 8; ++e) a[i][e] = rnd_bf16(threadIdx.x * 131 + i * 17 + e, a0[e]); }
  for (int j = 0; j < 2; ++j) { b[j] = b0; for (int e = 0; e < 8; ++e) b[j][e] = rnd_bf16(threadIdx.x * 257 + j * 29 + e + 7, b0[e]); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i * 2 + j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4 * 8);
  bf16x8 a = {0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00}, b = a;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512}) {
    const int iters = 40000;
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, d, iters, a, b);
        else hipLaunchKernelGGL(k32, dim3(blocks), dim3(512), 0, 0, d, iters, a, b);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 8 * iters * 32 * 2.0 * 16 * 16 * 32;   // both kernels: 32 x 16x16x32-equivalents per iter
        if (rep) printf("%s blocks=%d: %.3f ms  %.0f TFLOP/s\n", which ? "32x32x16" : "16x16x32", blocks, ms, fl / ms / 1e9);
      }
    }
  }
  return 0;
}
