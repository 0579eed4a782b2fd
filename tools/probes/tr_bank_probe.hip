// Timing probe for ds_read_b64_tr_b16 bank behaviour on gfx950 (dev tool).  One wave, 32 back-to-back reads per iteration at
// lane addresses from a host table (+ a wave-uniform offset per read); prints cycles per read for the transposing read and
// for a plain ds_read_b64 at the same addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 lds_s4;
template <bool TR>
__global__ __launch_bounds__(512) void probe(const int* addr, long* out, int iters, int step) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((int*)lds)[i] = i;
  __syncthreads();
  const int a = addr[threadIdx.x & 63];
  s4 acc = {0, 0, 0, 0};
  long t0 = 0, t1 = 0;
  for (int w = 0; w < 2; ++w) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      s4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        lds_s4* p = (lds_s4*)((__attribute__((address_space(3))) char*)lds + ((a + j * step + (it & 3) * 16384 + (threadIdx.x >> 6) * 2048 * 16) & 65535 & ~7));
        if (TR) v[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
        else v[j] = *p;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    t1 = __builtin_readcyclecounter();
  }
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (acc[0] == 12345 && acc[1] == 777) out[1] = acc[2];
}
int main() {
  int* d_addr; long* d_out; int h[64]; long ho[2];
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 16);
  hipFuncSetAttribute((const void*)probe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)probe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const char* names[] = {"uniform", "linear 8B/lane", "A_tn 128B rows swz", "A_tn 128B rows no swz", "B_tn 64B rows swz", "B_tn 64B rows no swz",
                         "guide 32B rows, groups 512B apart", "attn 256B rows chunk^row", "32B rows, groups 128B apart (fully linear rows)",
                         "128B rows, swz by row&3", "128B rows, swz by (row>>2)", "64B rows, swz by row&1", "256B rows no swz",
                         "128B rows: lanes b spread 32B units", "64B rows swz (row>>1)&1"};
  for (int pat = 0; pat < 15; ++pat) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, a = (l & 15) >> 2, b = l & 3, row = 4 * g + a;
      switch (pat) {
        case 0: h[l] = 0; break;
        case 1: h[l] = 8 * l; break;
        case 2: h[l] = 128 * row + 32 * ((row >> 1) & 3) + 8 * b; break;
        case 3: h[l] = 128 * row + 8 * b; break;
        case 4: h[l] = 64 * row + 32 * ((row >> 2) & 1) + 8 * b; break;
        case 5: h[l] = 64 * row + 8 * b; break;
        case 6: h[l] = 8 * ((l & 15) + g * 64); break;
        case 7: h[l] = 256 * row + (((b >> 1) ^ row) << 4) + (b & 1) * 8; break;
        case 8: h[l] = 32 * row + 8 * b; break;
        case 9: h[l] = 128 * row + 32 * (row & 3) + 8 * b; break;
        case 10: h[l] = 128 * row + 32 * ((row >> 2) & 3) + 8 * b; break;
        case 11: h[l] = 64 * row + 32 * (row & 1) + 8 * b; break;
        case 12: h[l] = 256 * row + 8 * b; break;
        case 13: h[l] = 128 * row + 32 * b; break;
        case 14: h[l] = 64 * row + 32 * ((row >> 1) & 1) + 8 * b; break;
      }
    }
    hipMemcpy(d_addr, h, 256, hipMemcpyHostToDevice);
    for (int waves = 1; waves <= 8; waves *= 8) {
    double cyc[2];
    for (int tr = 0; tr < 2; ++tr) {
      const int iters = 2000;
      if (tr) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64 * waves), 65536, 0, d_addr, d_out, iters, 2048);
      else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64 * waves), 65536, 0, d_addr, d_out, iters, 2048);
      hipMemcpy(ho, d_out, 16, hipMemcpyDeviceToHost);
      cyc[tr] = (double)ho[0] / (iters * 16.0);
    }
    printf("waves %d pat %2d %-48s plain b64 %.2f  tr_b16 %.2f  (counter ticks per read per wave)\n", waves, pat, names[pat], cyc[0], cyc[1]);
  }
  }
  return 0;
}
