// Dev probe: what one wave per SIMD pays per instruction KIND placed between 32x32x16 MFMAs (two alternating accumulator chains in VGPRs, 4 waves per
// block, 256 blocks).  Variants add, per MFMA: VALU fillers, an LDS read + counted wait, transposing reads, and per 8 MFMAs an LDS-DMA piece / a barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_mix_probe.hip -o /tmp/issue_mix_probe && /tmp/issue_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int NV, int NLDS, int NTR, int DMA, int BAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(long* out, const unsigned short* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  asm volatile("" ::: "v255", "a255");
  const uint32_t lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  uint32_t addr = lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  asm volatile("" : "+v"(addr));
  // buffer descriptor over src
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint64_t a = (uint64_t)(uintptr_t)src;
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((uint32_t)a); rs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu); rs[2] = 0x7fffffff; rs[3] = 0x00020000u;
  uint32_t voff = (threadIdx.x & 63) * 16 + blockIdx.x * 65536;
  const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds + 32768 + (threadIdx.x >> 6) * 1024);
  asm volatile("" : "+v"(voff));
  asm volatile("s_nop 4");
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (BAR && u == 8) asm volatile("s_barrier" ::: "memory");
      if (DMA && (u & 7) == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rs) : "memory");
      if (DMA && (u & 7) == 7) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (NLDS) { asm volatile("ds_read_b128 v[216:219], %0 offset:0" ::"v"(addr)); asm volatile("s_waitcnt lgkmcnt(3)"); }
      if (NTR) { asm volatile("ds_read_b64_tr_b16 v[220:221], %0 offset:8192" ::"v"(addr)); asm volatile("ds_read_b64_tr_b16 v[222:223], %0 offset:12288" ::"v"(addr)); asm volatile("s_waitcnt lgkmcnt(3)"); }
      if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 v[64:79], v[200:203], v[204:207], v[64:79]");
      else asm volatile("v_mfma_f32_32x32x16_bf16 v[80:95], v[200:203], v[204:207], v[80:95]");
      if (NV >= 2 && NV < 100 || NV >= 100) { asm volatile("v_mul_f32 v210, v210, v211"); asm volatile("v_exp_f32 v212, v212"); }
      if (NV >= 4 && NV < 100) { asm volatile("v_mul_f32 v213, v213, v211"); asm volatile("v_exp_f32 v214, v214"); }
      if (NV >= 6 && NV < 100) { asm volatile("v_cvt_pk_bf16_f32 v215, v210, v213"); asm volatile("v_fma_f32 v224, v224, v211, v225"); }
      if (NV == 7) { asm volatile("v_max3_f32 v226, v226, v210, v213"); }
      if (NV == 102) { asm volatile("v_pk_mul_f32 v[228:229], v[228:229], v[230:231]"); asm volatile("v_pk_fma_f32 v[232:233], v[232:233], v[230:231], v[234:235]"); }
      if (NV == 104) { asm volatile("v_pk_mul_f32 v[228:229], v[228:229], v[230:231]"); asm volatile("v_pk_fma_f32 v[232:233], v[232:233], v[230:231], v[234:235]");
                       asm volatile("v_pk_mul_f32 v[236:237], v[236:237], v[230:231]"); asm volatile("v_pk_add_f32 v[238:239], v[238:239], v[230:231]"); }
      if (NV == 202) { asm volatile("s_waitcnt lgkmcnt(3)"); asm volatile("s_waitcnt lgkmcnt(3)"); }
      if (NV == 204) { asm volatile("s_waitcnt lgkmcnt(3)"); asm volatile("s_waitcnt lgkmcnt(3)"); asm volatile("s_waitcnt lgkmcnt(3)"); asm volatile("s_waitcnt lgkmcnt(3)"); }
      if (NV == 304) { asm volatile("s_add_u32 s40, s40, 4\n\ts_add_u32 s41, s41, 4\n\ts_add_u32 s42, s42, 4\n\ts_add_u32 s43, s43, 4" ::: "s40", "s41", "s42", "s43", "scc"); }
      if (NV == 308) { asm volatile("s_add_u32 s40, s40, 4\n\ts_add_u32 s41, s41, 4\n\ts_add_u32 s42, s42, 4\n\ts_add_u32 s43, s43, 4\n\ts_add_u32 s40, s40, 4\n\ts_add_u32 s41, s41, 4\n\ts_add_u32 s42, s42, 4\n\ts_add_u32 s43, s43, 4" ::: "s40", "s41", "s42", "s43", "scc"); }
      if (NV == 402) { asm volatile("ds_read_b128 v[216:219], %0 offset:0" ::"v"(addr)); asm volatile("ds_read_b128 v[240:243], %0 offset:4096" ::"v"(addr)); }
      if (NV == 404) { asm volatile("ds_read_b128 v[216:219], %0 offset:0" ::"v"(addr)); asm volatile("ds_read_b128 v[240:243], %0 offset:4096" ::"v"(addr));
                       asm volatile("ds_read_b128 v[244:247], %0 offset:8192" ::"v"(addr)); asm volatile("ds_read_b128 v[248:251], %0 offset:12288" ::"v"(addr)); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NV, int NLDS, int NTR, int DMA, int BAR>
void run(long* d, const unsigned short* src, const char* what) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<NV, NLDS, NTR, DMA, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV, NLDS, NTR, DMA, BAR>), dim3(256), dim3(256), 65536, 0, d, src, iters);
  hipDeviceSynchronize();
  long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-78s %.1f clock64 ticks per MFMA\n", what, (double)h / (iters * 16.0));
}
int main() {
  long* d; unsigned short* src;
  hipMalloc(&d, 64); hipMalloc(&src, 256 * 65536 + 65536); hipMemset(src, 0, 256 * 65536 + 65536);
  run<0, 0, 0, 0, 0>(d, src, "MFMAs only");
  run<4, 0, 0, 0, 0>(d, src, "+ 4 VALU (2 v_exp) per MFMA");
  run<6, 0, 0, 0, 0>(d, src, "+ 6 VALU per MFMA");
  run<7, 0, 0, 0, 0>(d, src, "+ 7 VALU per MFMA");
  run<102, 0, 0, 0, 0>(d, src, "+ 2 VALU + 2 packed fp32 (pk_mul, pk_fma) per MFMA");
  run<104, 0, 0, 0, 0>(d, src, "+ 2 VALU + 4 packed fp32 per MFMA");
  run<202, 0, 0, 0, 0>(d, src, "+ 2 VALU + 2 s_waitcnt per MFMA");
  run<204, 0, 0, 0, 0>(d, src, "+ 2 VALU + 4 s_waitcnt per MFMA");
  run<304, 0, 0, 0, 0>(d, src, "+ 2 VALU + 4 SALU per MFMA");
  run<308, 0, 0, 0, 0>(d, src, "+ 2 VALU + 8 SALU per MFMA");
  run<402, 0, 0, 0, 0>(d, src, "+ 2 VALU + 2 ds_read_b128 (no wait) per MFMA");
  run<404, 0, 0, 0, 0>(d, src, "+ 2 VALU + 4 ds_read_b128 (no wait) per MFMA");
  run<0, 1, 0, 0, 0>(d, src, "+ 1 ds_read_b128 + counted wait per MFMA");
  run<0, 0, 1, 0, 0>(d, src, "+ 2 ds_read_b64_tr_b16 + counted wait per MFMA");
  run<4, 1, 0, 0, 0>(d, src, "+ 4 VALU + 1 ds_read_b128 per MFMA");
  run<4, 0, 1, 0, 0>(d, src, "+ 4 VALU + 2 transposing reads per MFMA");
  run<0, 0, 0, 1, 0>(d, src, "+ 1 LDS-DMA piece per 8 MFMAs");
  run<0, 0, 0, 0, 1>(d, src, "+ 1 barrier per 16 MFMAs");
  run<4, 1, 0, 1, 1>(d, src, "+ 4 VALU + ds_read_b128 + DMA piece / 8 + barrier / 16");
  run<6, 0, 1, 1, 1>(d, src, "+ 6 VALU + 2 transposing reads + DMA piece / 8 + barrier / 16");
  return 0;
}
