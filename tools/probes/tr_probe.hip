// Probe of ds_read_b64_tr_b16 semantics on gfx950 (dev tool). LDS[i] = i (u16). Each lane supplies a byte
// address; prints which LDS element indices each lane receives.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void probe(const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (uint16_t)i;
  __syncthreads();
  int a = addr[threadIdx.x];
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)((__attribute__((address_space(3))) char*)lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  int* d_addr; uint16_t* d_out; int h_addr[64]; uint16_t h_out[256];
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
  for (int pat = 0; pat < 5; ++pat) {
    for (int l = 0; l < 64; ++l) {
      int i = l & 15, g = l >> 4;
      switch (pat) {
        case 0: h_addr[l] = 0; break;                               // uniform
        case 1: h_addr[l] = l * 8; break;                           // contiguous 8 B per lane
        case 2: h_addr[l] = i * 200 + g * 8; break;                 // lane i -> row i (stride 100 el), group g -> +4 el
        case 3: h_addr[l] = (i >> 2) * 200 + (i & 3) * 8 + g * 1000; break;   // 4 rows x 16 cols block per group, row stride 100 el
        case 4: h_addr[l] = (i & 3) * 200 + (i >> 2) * 8 + g * 1000; break;   // transposed assignment
      }
    }
    hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) {
      printf(" l%02d a=%5d(el %4d): %4d %4d %4d %4d", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
      if (l % 2 == 1) printf("\n");
    }
  }
  return 0;
}
