// What does the lane -> address pattern of a 16-byte vector store / load cost on one CU's memory path?  (dev probe behind the GEMM epilogue question:
// the 4-wave kernel's register-direct epilogue writes 16 rows x 64 contiguous bytes per buffer_store_dwordx4 and measures ~70 cycles per
// instruction and CU.)  256 blocks x 4 waves (one per SIMD, like gemm_nt_256w4), every wave moves its 128 x 128 bf16 quarter of a 256 x 256
// tile of a [16384 x LD] matrix, tile after tile (v += 256), with one of four patterns per 1 KB instruction:
//   0: 16 rows x 64 B   (the epilogue's: lane (fr, g) -> row fr, 16 B at 16 g; two instructions fill a row's 128-byte line)
//   1:  8 rows x 128 B  (lane l -> row l >> 3, chunk l & 7)
//   2:  4 rows x 256 B  (lane l -> row l >> 4, chunk l & 15: the wave's whole row)
//   3: 16 rows x 4 x 16 B at a 32-byte stride (the SwiGLU backward's gate / up loads: half of every line per instruction)
// build: hipcc --offload-arch=gfx950 -O3 store_pattern_probe.hip -o /tmp/spp ; run: /tmp/spp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int PAT, bool LOAD>
__global__ __launch_bounds__(256) void probe(uint16_t* C, int ld, int tiles_n, int ntiles, u32x4* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  u32x4 acc = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const int m0 = (v / tiles_n) * 256 + wr * 128, n0 = (v % tiles_n) * 256 + wc * 128;
    uint16_t* base = C + (long)m0 * ld + n0;
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      int row, colb;      // row within the 128-row quarter, byte column within its 256-byte row
      if (PAT == 0) { row = (it >> 2) * 16 + (lane & 15); colb = (it & 3) * 64 + (lane >> 4) * 16; }
      else if (PAT == 1) { row = (it >> 1) * 8 + (lane >> 3); colb = (it & 1) * 128 + (lane & 7) * 16; }
      else if (PAT == 2) { row = it * 4 + (lane >> 4); colb = (lane & 15) * 16; }
      else { row = (it >> 2) * 16 + (lane & 15); colb = (it & 1) * 16 + (it & 2) * 64 + (lane >> 4) * 32; }
      u32x4* ptr = (u32x4*)((char*)(base + (long)row * ld) + colb);
      if (LOAD) { const u32x4 x = __builtin_nontemporal_load(ptr); acc += x; }
      else *ptr = acc;
    }
  }
  if (LOAD && acc[0] == 0x12345678u) sink[threadIdx.x] = acc;
}
template <int PAT, bool LOAD>
static float run(uint16_t* C, int ld, int tiles_n, int ntiles, u32x4* sink, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<PAT, LOAD>), dim3(grid), dim3(256), 0, 0, C, ld, tiles_n, ntiles, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<PAT, LOAD>), dim3(grid), dim3(256), 0, 0, C, ld, tiles_n, ntiles, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10 * 1e3f;
}
int main() {
  const int M = 16384, LD = 28672, tiles_n = LD / 256, ntiles = (M / 256) * tiles_n;      // 7168 tiles = 940 MB
  uint16_t* C; hipMalloc(&C, (size_t)M * LD * 2); hipMemset(C, 1, (size_t)M * LD * 2);
  u32x4* sink; hipMalloc(&sink, 4096);
  for (int grid : {256, 32}) {
    const int nt = grid == 256 ? ntiles : ntiles / 8;
    const double mb = (double)nt * 131072 / 1e6, per_cu_tile = 131072.0;
    float t;
#define REP(P, L, NAME)                                                                                                   \
    t = run<P, L>(C, LD, tiles_n, nt, sink, grid);                                                                        \
    printf("grid %3d %-34s %8.1f us  %6.2f TB/s  %5.2f us per tile and CU (%4.1f B/clk/CU at 2.1 GHz)\n", grid, NAME, t, mb / t, \
           t / (nt / (double)grid), per_cu_tile / (t / (nt / (double)grid) * 2100.0));
    REP(0, false, "store 16 rows x 64 B");
    REP(1, false, "store  8 rows x 128 B");
    REP(2, false, "store  4 rows x 256 B");
    REP(3, false, "store 16 rows x 4x16 B / 32 B");
    REP(0, true, "load  16 rows x 64 B");
    REP(1, true, "load   8 rows x 128 B");
    REP(2, true, "load   4 rows x 256 B");
    REP(3, true, "load  16 rows x 4x16 B / 32 B");
  }
  return 0;
}
