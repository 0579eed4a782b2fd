// Does the raw-buffer range check of gfx950 include the scalar offset?  (dev probe: the attention kernels put a tile's row offset into soffset and
// rely on out-of-range rows reading as zeros.)  Buffer of 64 dwords, num_records = 64 * 4 bytes; every lane loads dword `lane` at soffset = s bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const int* buf, int* out, int soff) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 64 * 4, 0x00020000);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, so, 0);
}
int main() {
  int h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int soff : {0, 128, 256, 512}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, soff);
    int r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("soffset %3d bytes: lane0 %d lane31 %d lane32 %d lane63 %d   (in-range data = 1000 + lane + soffset/4; 0 = range-checked)\n", soff, r[0], r[31], r[32], r[63]);
  }
  return 0;
}
