// bf16 MFMA GEMM for the VisPer-LM hot path:  C[M,N] = epi(A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
// Both operands are K-contiguous ("NT"): activations [tokens, features] x nn.Linear weights [out, in].
// (dgrad uses the pre-transposed frozen weight, wgrad uses explicitly transposed operands — DESIGN.md.)
//
// Kernels (dispatch in vp_gemm_bf16):
//   gemm_nt_256p8   the production kernel for every large problem: persistent 256x256x64 tiles, 8 waves, 8-phase ping-pong
//                   (see the comment on the kernel), fused bias / activation / residual / SwiGLU epilogues
//   gemm_nt_256     the plain persistent 256-tile kernel it grew out of (one barrier per K-tile); kept as the A/B reference
//   gemm_nt_128     128x128x64, 4 waves (2x2, each 64x64): small problems (heads, ViT N=1024, DPT convs)
//   gemm_nt_generic bounds-checked fallback for K % 64 != 0 or unaligned rows
// Common ground: operands go HBM -> LDS with 16-byte global_load_lds (no VGPR round trip).  The LDS image is lane-linear (a
// glds constraint), so bank conflicts are removed by XOR-swizzling the 16-B chunk index on the *source* address and again on the
// ds_read_b128 (chunk ^= (row>>1)&7).  MFMA roles are swapped (A-operand = weight rows, B-operand = token rows) so each lane
// ends up with 4 consecutive output columns of one token row.  Block ids are remapped XCD-aware so tiles sharing an operand
// panel sit in one XCD's L2.
#include "common.h"
#include <cstdlib>
#include <utility>
#include <type_traits>

// 4-wave GEMM kernel: gap (of K-step 1) -> which of the 16 fragment reads of the next K-tile's K-step 0 sits there (-1: none)
constexpr int vp_w4_rd_slot(int g) {
  constexpr int RG[16] = {13, 15, 18, 20, 22, 25, 27, 29, 32, 34, 36, 39, 41, 43, 46, 48};      // never a DMA (16 + 7k) / m0 (17 + 7k) / offset (19 + 7k) gap
  for (int r = 0; r < 16; ++r)
    if (RG[r] == g) return r;
  return -1;
}
// byte offset of B fragment r = 4 h + 2 s + jj from the lane's base row: rows 64 h + 32 s + 4 jj of 128 bytes
constexpr int vp_w4_boff(int r) { return ((r >> 2) * 64 + ((r >> 1) & 1) * 32 + (r & 1) * 4) * 128; }

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_QUICK_GELU = 2, EPI_RELU = 3 };

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* res;
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int epi;
  int dbg;                                             // experiments only (VP_GEMM_DBG), 0 in production
  // fused SwiGLU epilogues (8-phase kernel, interior tiles only; gate/up columns interleaved in 8-wide chunks: g8 | u8 | ...):
  //   mode 1: C[M,N] = gate_up (as usual) and C2[M,N/2] = silu(gate) * up
  //   mode 2: the accumulator is d_act[M,N]; aux = gate_up[M,2N]; C[M,2N] = d_gate_up
  int mode;
  void* C2;
  long ldc2;
  const bf16_t* aux;
  long ldaux;
  // dynamic tile scheduling (8-phase kernel, persistent grid only): 8 per-XCD claim counters + a done counter, or NULL = static v += 256
  int* sched;
  // bf16 C tiles leave with non-temporal stores (set by the launcher for N <= 8192, the shapes where it measures +1...2 %: the output does not
  // evict the operand panels from the XCD's L2; hipBLASLt's kernels store C the same way.  Wider outputs measured -0.7 %.)
  int c_nt;
  // general variant of the 4-wave kernel only (gemm_nt_256w4<false, 1>): optional fp32 per-row scale applied to the accumulator before the bias
  // (RMSNorm's 1/rms when gamma is folded into the frozen weight, HF LlamaRMSNorm: modeling_llama.py)
  const float* rowscale;
  // general variant only: rotate-half RoPE (head_dim 128 = one wave sub-tile) on the columns < rope_cols of the bf16-rounded result, with the
  // rounding points of rope_kernel (elementwise.hip); cos / sin fp32 [S, 64], position of row r = rope_pos ? rope_pos[r] : r % rope_S
  // lean variant, residual epilogue only: per-row sum of squares of the stored row (fp32, before the last bf16 rounding) over this wave's 128 columns,
  // written to sumsq_part[row * (N / 16) + 16-column block]: the next RMSNorm's statistics without another pass over the residual stream
  float* sumsq_part;
  const float* rope_cos;
  const float* rope_sin;
  const int* rope_pos;
  int rope_S, rope_cols;
};

__device__ __forceinline__ float apply_epi(float v, int epi) {
  switch (epi) {
    case EPI_GELU: return gelu_erf(v);
    case EPI_QUICK_GELU: return quick_gelu(v);
    case EPI_RELU: return fmaxf(v, 0.f);
    default: return v;
  }
}

template <int EPI>
__device__ __forceinline__ float apply_epi_c(float v) {
  if constexpr (EPI == EPI_GELU) return gelu_erf(v);
  else if constexpr (EPI == EPI_QUICK_GELU) return quick_gelu(v);
  else if constexpr (EPI == EPI_RELU) return fmaxf(v, 0.f);
  else return v;
}

// Store 4 consecutive columns (n..n+3) of row m with the reference's rounding points:
// linear(+bias) -> bf16, activation -> bf16, residual add -> bf16 (HF bf16 modules round after each op).
template <bool OUT_F32>
__device__ __forceinline__ void store4(const GemmArgs& p, int m, int n, f32x4 acc) {
  if (m >= p.M || n >= p.N) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const int nv = min(4, p.N - n);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r < nv) {
      float x = v[r];
      if (p.bias) x += bf2f(p.bias[n + r]);
      if (!OUT_F32) x = bfround(x);
      if ((p.epi & 0xff) != EPI_NONE) {
        x = apply_epi(x, p.epi & 0xff);
        if (!OUT_F32) x = bfround(x);
      }
      if (p.res) {
        x += bf2f(p.res[(long)m * p.ldr + n + r]);
      }
      v[r] = x;
    }
  }
  if (OUT_F32) {
    float* c = (float*)p.C + (long)m * p.ldc + n;
    if (nv == 4 && ((((uintptr_t)c) & 15) == 0)) {
      *(f32x4*)c = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      for (int r = 0; r < nv; ++r) c[r] = v[r];
    }
  } else {
    bf16_t* c = (bf16_t*)p.C + (long)m * p.ldc + n;
    if (nv == 4 && ((((uintptr_t)c) & 7) == 0)) {
      bf16x4 o;
      o[0] = (short)f2bf(v[0]); o[1] = (short)f2bf(v[1]); o[2] = (short)f2bf(v[2]); o[3] = (short)f2bf(v[3]);
      *(bf16x4*)c = o;
    } else {
      for (int r = 0; r < nv; ++r) c[r] = f2bf(v[r]);
    }
  }
}


#define GLDS16(gptr, ldsptr)                                                                          \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),             \
                                   (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

// ------------------------------------------------------------------------------------------------
// fast path: K % 64 == 0, 16-byte aligned rows
// ------------------------------------------------------------------------------------------------
template <int HALVES>
__device__ __forceinline__ void epilogue_swz(const GemmArgs& p, bf16_t* wave_lds, const f32x4 (&acc)[HALVES * 4][4], int mrow0, int ncol0,
                                            int lane);      // defined below (shared LDS-staged epilogue)

template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_128(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 128 * 64];
  bf16_t* As = smem;
  bf16_t* Bs = smem + 128 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int tiles_m = (p.M + 127) >> 7, tiles_n = (p.N + 127) >> 7;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP_M = 8;
  const int width = GROUP_M * tiles_n;
  const int group = id / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (id % width) % gsz;
  const int tn = (id % width) / gsz;
  const int m0 = tm << 7, n0 = tn << 7;

  // staging: 1024 16-B chunks per operand tile, 4 per thread; LDS chunk q = (row q>>3, slot q&7) holds
  // global chunk (q&7) ^ ((row>>1)&7) of that row.
  const bf16_t* srcA[4];
  const bf16_t* srcB[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = it * 256 + tid;
    const int row = q >> 3;
    const int gc = (q & 7) ^ ((row >> 1) & 7);
    const int ar = min(m0 + row, p.M - 1);
    const int br = min(n0 + row, p.N - 1);
    srcA[it] = p.A + (long)ar * p.lda + gc * 8;
    srcB[it] = p.B + (long)br * p.ldb + gc * 8;
  }

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += 64) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      GLDS16(srcA[it] + k0, As + (it * 256 + wave * 64) * 8);
      GLDS16(srcB[it] + k0, Bs + (it * 256 + wave * 64) * 8);
    }
    __syncthreads();   // LDS-DMA pending -> the compiler's barrier carries vmcnt(0)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[4], wf[4];
      const int cg = ks * 4 + g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        xf[i] = *(const bf16x8*)(As + r * 64 + ((cg ^ ((r >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + fr;
        wf[j] = *(const bf16x8*)(Bs + r * 64 + ((cg ^ ((r >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // D tile (A-operand = weight rows): row index = output column n (4g + r), col index = token row m (fr)
  if (!OUT_F32) {
    // bf16 output: the operand tiles are dead (last barrier above) -> each wave stages its 64 x 64 sub-tile in its own 8 KB slice and
    // writes 16-byte coalesced rows with the fused bias / activation / residual (same lean path as the 256-tile kernels)
    epilogue_swz<1>(p, smem + wave * 4096, acc, m0 + wm * 64, n0 + wn * 64, lane);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      store4<OUT_F32>(p, m, n, acc[i][j]);
    }
}

// ------------------------------------------------------------------------------------------------
// large-problem path: PERSISTENT 256x256x64 kernel.  One 512-thread block per CU (8 waves = 2 x 4, each
// 128x64 = 8x4 MFMA tiles), 2 LDS buffers (2 x 64 KB).  The K-tile stream never stops: tile t+1 is DMA'd
// (global_load_lds) into the other buffer at the START of computing tile t, and at a block's LAST K-tile the
// DMA already fetches the first K-tile of the block's NEXT output tile, so there is no prologue bubble between
// output tiles, the epilogue's stores drain under the next tile's MFMAs, and the end-of-tile write bursts of
// the 256 CUs stop being synchronised.  (Ablation: with one block per output tile the turnover — exposed first
// load + store drain + relaunch — cost ~28 % of the kernel at K = 4096.)  The epilogue stages C through the LDS
// buffer that was just consumed (XOR-swizzled 128-byte rows) to emit 16-byte, 128-byte-row-coalesced accesses.
// ------------------------------------------------------------------------------------------------
struct TileCoord {
  int m0, n0;
};
__device__ __forceinline__ TileCoord tile_coord_256(int v, int tiles_m, int tiles_n) {
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(v, nwg);
  const int GROUP_M = 8;
  const int width = GROUP_M * tiles_n;
  const int group = id / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  return TileCoord{(first_m + (id % width) % gsz) << 8, ((id % width) / gsz) << 8};
}

// Persistent 256-block grids: step s = v / 256 walks super-blocks of SBM x SBN tiles (SBM * SBN = 256, all concurrently
// resident), XCD x = v & 7 owns an 8 x 4 sub-block of it, so one K-step touches SBM + SBN operand panels chip-wide (32 for
// 16 x 16) instead of 64 + 4 with the row-group walk above, and 12 per XCD L2.
__device__ __forceinline__ TileCoord tile_coord_sb(int v, int tiles_m, int tiles_n, int sbm, int sbn) {
  const int s = v >> 8, b = v & 255, x = b & 7, slot = b >> 3;
  const int sb_per_row = tiles_n / sbn;                // super-blocks along N
  const int sbr = s / sb_per_row, sbc = s - sbr * sb_per_row;
  const int subs_n = sbn >> 2;                         // 8x4 sub-blocks along N inside a super-block
  const int xr = x / subs_n, xc = x - xr * subs_n;
  return TileCoord{(sbr * sbm + xr * 8 + (slot & 7)) << 8, (sbc * sbn + xc * 4 + (slot >> 3)) << 8};
}

// swizzled C staging: per wave a [64 rows][64 cols] bf16 slice (8 KB), 16-byte chunk index ^= row & 7
__device__ long vp_dbg_stamps[256 * 8];
template <int HALVES>      // 64-row halves of the wave's sub-tile: 2 for the 256-tile kernels (128 x 64 per wave), 1 for the 128-tile kernel
__device__ __forceinline__ void epilogue_swz(const GemmArgs& p, bf16_t* wave_lds, const f32x4 (&acc)[HALVES * 4][4], int mrow0, int ncol0,
                                            int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const int epi = p.epi & 0xff;
  // Fast path: interior tile, 16-byte aligned rows (every decoder GEMM; with bias / activation: ViT, heads, DPT).  Kept lean: the
  // general path below is ~10x the instructions, and the epilogue runs with the matrix pipe idle.
  const bool fast = __builtin_amdgcn_readfirstlane(
      (int)(mrow0 + 64 * HALVES <= p.M && ncol0 + 64 <= p.N && (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0 &&
            (!p.bias || (((uintptr_t)p.bias) & 7) == 0) &&
            (!p.res || ((p.ldr & 7) == 0 && (((uintptr_t)p.res) & 15) == 0))));
  if (fast) {
    int woff[4];                                        // this lane's write offset per column block j (row term added per ii)
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = fr * 64 + (((j * 2 + (g >> 1)) ^ (fr & 7)) << 3) + (g & 1) * 4;
    const int rl0 = lane >> 3, ch = lane & 7;
    const int roff = rl0 * 64 + ((ch ^ (rl0 & 7)) << 3); // read offset for it = 0; it adds 8 rows (same swizzle phase)
    bf16_t* cptr = (bf16_t*)p.C + (long)(mrow0 + rl0) * p.ldc + ncol0 + ch * 8;
    const bf16_t* rptr = p.res ? p.res + (long)(mrow0 + rl0) * p.ldr + ncol0 + ch * 8 : nullptr;
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
      if (!p.bias && epi == EPI_NONE) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 a = acc[half * 4 + ii][j];
            u32x2 o;
            o[0] = pack_bf16x2(a[0], a[1]);
            o[1] = pack_bf16x2(a[2], a[3]);
            *(u32x2*)(wave_lds + ii * 16 * 64 + woff[j]) = o;
          }
      } else {                                          // bias and/or activation (ViT, heads, DPT decoder): same rounding points
        float bias4[4][4];                              // as the general path: linear(+bias) -> bf16, activation -> bf16
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x4 bv = {0, 0, 0, 0};
          if (p.bias) bv = *(const bf16x4*)(p.bias + ncol0 + j * 16 + g * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) bias4[j][r] = bf2f((bf16_t)bv[r]);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 a = acc[half * 4 + ii][j];
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              x[r] = a[r] + bias4[j][r];
              if (epi != EPI_NONE) x[r] = apply_epi(bfround(x[r]), epi);
            }
            u32x2 o;
            o[0] = pack_bf16x2(x[0], x[1]);
            o[1] = pack_bf16x2(x[2], x[3]);
            *(u32x2*)(wave_lds + ii * 16 * 64 + woff[j]) = o;
          }
      }
      __builtin_amdgcn_sched_barrier(0);                // residual loads only after this half's accumulators are dead
      if (p.mode == 2) {
        // d_act tile is staged (bf16-rounded, as the unfused path stores it); lane (row rl0 + 8 it, chunk ch) owns 8 d_act
        // columns == one (g8 | u8) chunk pair of gate_up / d_gate_up
        const bf16_t* gptr = p.aux + (long)(mrow0 + half * 64 + rl0) * p.ldaux + 2 * (ncol0 + ch * 8);
        bf16_t* dptr = (bf16_t*)p.C + (long)(mrow0 + half * 64 + rl0) * p.ldc + 2 * (ncol0 + ch * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
          bf16x8 gv[4], uv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            gv[k] = *(const bf16x8*)(gptr + (long)((grp * 4 + k) * 8) * p.ldaux);
            uv[k] = *(const bf16x8*)(gptr + (long)((grp * 4 + k) * 8) * p.ldaux + 8);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bf16x8 dv = *(const bf16x8*)(wave_lds + (grp * 4 + k) * 8 * 64 + roff);
            bf16x8 og, ou;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gg = bf2f((bf16_t)gv[k][e]), uu = bf2f((bf16_t)uv[k][e]), dd = bf2f((bf16_t)dv[e]);
              const float sg = 1.f / (1.f + __expf(-gg));
              og[e] = (short)f2bf(dd * uu * sg * (1.f + gg * (1.f - sg)));
              ou[e] = (short)f2bf(dd * gg * sg);
            }
            *(bf16x8*)(dptr + (long)((grp * 4 + k) * 8) * p.ldc) = og;
            *(bf16x8*)(dptr + (long)((grp * 4 + k) * 8) * p.ldc + 8) = ou;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        continue;
      }
      u32x4 rv[8];
      if (rptr) {
#pragma unroll
        for (int it = 0; it < 8; ++it) rv[it] = *(const u32x4*)(rptr + (long)(half * 64 + it * 8) * p.ldr);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        u32x4 v = *(const u32x4*)(wave_lds + it * 8 * 64 + roff);
        if (rptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __builtin_bit_cast(float, v[e] << 16) + __builtin_bit_cast(float, rv[it][e] << 16);
            const float hi = __builtin_bit_cast(float, v[e] & 0xffff0000u) + __builtin_bit_cast(float, rv[it][e] & 0xffff0000u);
            v[e] = pack_bf16x2(lo, hi);
          }
        }
        if (p.c_nt) {
          // non-temporal store (cache policy nt = aux 2): the C tile does not linger in the XCD's L2, which the operand panels re-use
          const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0x7fffffff, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(v, crs, (int)((const char*)cptr - (const char*)p.C),
                                                 (int)((long)(half * 64 + it * 8) * p.ldc * 2), 2);
        } else {
          *(u32x4*)(cptr + (long)(half * 64 + it * 8) * p.ldc) = v;
        }
      }
      if (p.mode == 1) {
        // act = silu(gate) * up from the staged gate_up tile: lane (row rl + 16 it, chunk pair pr) -> 8 act columns
        const int rl = lane >> 2, pr = lane & 3;
        bf16_t* aptr = (bf16_t*)p.C2 + (long)(mrow0 + half * 64 + rl) * p.ldc2 + (ncol0 >> 1) + pr * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 16 + rl;
          const bf16x8 gv = *(const bf16x8*)(wave_lds + row * 64 + (((2 * pr) ^ (row & 7)) << 3));
          const bf16x8 uv = *(const bf16x8*)(wave_lds + row * 64 + (((2 * pr + 1) ^ (row & 7)) << 3));
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bfround(silu(bf2f((bf16_t)gv[e]))) * bf2f((bf16_t)uv[e]));
          *(bf16x8*)(aptr + (long)(it * 16) * p.ldc2) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  float bias4[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = ncol0 + j * 16 + g * 4 + r;
      bias4[j][r] = (p.bias && n < p.N) ? bf2f(p.bias[n]) : 0.f;
    }
#pragma unroll
  for (int half = 0; half < HALVES; ++half) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = half * 4 + ii;
      const int row = ii * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = bfround(acc[i][j][r] + bias4[j][r]);
          if (epi != EPI_NONE) x = bfround(apply_epi(x, epi));
          o[r] = (short)f2bf(x);
        }
        const int chunk = (j * 2 + (g >> 1)) ^ (row & 7);
        *(bf16x4*)(wave_lds + row * 64 + chunk * 8 + (g & 1) * 4) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private slice: own writes visible to own reads
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rl = it * 8 + (lane >> 3), ch = lane & 7;
      const int m = mrow0 + half * 64 + rl, n = ncol0 + ch * 8;
      bf16x8 v = *(const bf16x8*)(wave_lds + rl * 64 + ((ch ^ (rl & 7)) << 3));
      if (m < p.M && n < p.N) {
        bf16_t* cptr = (bf16_t*)p.C + (long)m * p.ldc + n;
        const bool full = (n + 8 <= p.N) && ((((uintptr_t)cptr) & 15) == 0);
        if (p.res) {
          const bf16_t* rptr = p.res + (long)m * p.ldr + n;
          if (full && ((((uintptr_t)rptr) & 15) == 0)) {
            const bf16x8 rv = *(const bf16x8*)rptr;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (short)f2bf(bf2f((bf16_t)v[e]) + bf2f((bf16_t)rv[e]));
          } else {
            for (int e = 0; e < 8 && n + e < p.N; ++e) v[e] = (short)f2bf(bf2f((bf16_t)v[e]) + bf2f(rptr[e]));
          }
        }
        if (full) *(bf16x8*)cptr = v;
        else for (int e = 0; e < 8 && n + e < p.N; ++e) cptr[e] = (bf16_t)v[e];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the slice is overwritten by the next pass
  }
}

// Whole-line stores from the MFMA accumulator layout.  x / y = the 16 bytes lane (fr, g) holds of the left / right 64-byte half of row fr's
// 128-byte line (16 rows per register).  z1 = what store 1 (rows 0..7 of the block, whole lines) takes from this lane, z2 = store 2 (rows 8..15):
//   lanes fr < 8:  z1 = own x,             z2 = y of lane fr + 8        lanes fr >= 8:  z1 = y of lane fr - 8,   z2 = own x
// One v_cndmask_b32_dpp (row_ror:8 on the y operand) per dword and store: 8 VALU per pair of stores.  (DPP reads of a VGPR written by the
// VALU instruction in front need two wait states: the s_nop.)
// Store-data keep-alive (round 6; the root cause of round 4's "NaNs beside other kernels", tools/nan_pattern_r06.py, EXPERIMENTS round 6).  The epilogues
// read the accumulators with `asm volatile("v_accvgpr_read_b32 ...")`, opaque to the compiler's hazard recogniser.  The register allocator hands the FIRST
// such read of the next 16-row block the VGPR that held dword 0 of the buffer_store_dwordx4 issued just before, and on gfx950 nothing holds a
// v_accvgpr_read back while a store still fetches its data from that VGPR: when the CU's memory pipeline is busy with ANOTHER kernel's waves (a
// co-resident wave of the other stream) the store picks its data up late and writes the next block's raw fp32 accumulator for lanes 12..15 of every
// 16-lane row (bit-exactly what the wrong elements were).  Alone on its CU the data is gone before the read issues, which is why the kernel was only ever
// wrong beside other kernels, and why claiming the whole register file (no foreign wave on the CU) hid it.  Fix: the previous block's store-data registers
// stay LIVE (an empty asm use) until this block's accumulator reads have issued, so the allocator cannot give them to those reads; their next writers are
// ordinary VALU instructions (cvt_pk / DPP), which the hardware does order behind the store's data fetch.
#ifndef VP_W4_STORE_KEEP
#define VP_W4_STORE_KEEP 1
#endif
#if VP_W4_STORE_KEEP
#define W4_KEEP2(A, B) asm volatile("" ::"v"(A), "v"(B))
#else
#define W4_KEEP2(A, B) do { } while (0)
#endif
__device__ __forceinline__ void w4_rows8_swap(u32x4& z1, u32x4& z2, const u32x4& x, const u32x4& y) {
  uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
  const unsigned long long lo8 = 0x00ff00ff00ff00ffull, hi8 = 0xff00ff00ff00ff00ull;
  asm volatile(
      "s_nop 1\n\t"
      "s_mov_b64 vcc, %[mlo]\n\t"
      "v_cndmask_b32_dpp %[a0], %[y0], %[x0], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a1], %[y1], %[x1], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a2], %[y2], %[x2], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a3], %[y3], %[x3], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[mhi]\n\t"
      "v_cndmask_b32_dpp %[b0], %[y0], %[x0], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b1], %[y1], %[x1], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b2], %[y2], %[x2], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b3], %[y3], %[x3], vcc row_ror:8 row_mask:0xf bank_mask:0xf"
      : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
      : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [y0] "v"(y[0]), [y1] "v"(y[1]), [y2] "v"(y[2]), [y3] "v"(y[3]),
        [mlo] "s"(lo8), [mhi] "s"(hi8)
      : "vcc");
  z1 = u32x4{a0, a1, a2, a3};
  z2 = u32x4{b0, b1, b2, b3};
}

// Epilogue of the 4-wave kernel (decoder GEMMs: no bias, no activation; optional residual; fused SwiGLU forward / backward): STRAIGHT from the
// accumulators to global memory, no LDS pass.  An MFMA result lane holds 4 consecutive columns of one token row; which weight row sits in which
// MFMA row of which B fragment is the K loop's free choice, and it reads them so that the four B fragments (2 s + jj) of a 64-column half h hand
// lane (fr, g) the columns 64 h + 32 s + 8 g + 4 jj + e: two runs of 8 consecutive columns per accumulator row = two 16-byte bf16 stores, the
// four lanes g of a row side by side (64 contiguous bytes per row and store instruction, the other half of the 128-byte line by the next one).
// Round 3's first version staged every 32 x 64 block through LDS as fp32 (8 passes of write / wait / read / wait: 4.5 us per tile).
// Rounding points as everywhere: accumulator -> bf16, (+ residual -> bf16) / (SwiGLU on bf16-rounded values): bit-identical to epilogue_swz.
// SwiGLU forward: the (g8 | u8) column pairs of gate_up put the gate block in the even-g lanes and the up block in the odd-g lanes; one
// v_permlane16_swap per dword (16-lane rows of two registers trade places) gives every lane a whole pair.
template <int EV = 0>      // 0: lean (decoder GEMMs) | 1: general (bias / activation / residual / RoPE, M tail) | 2: lean + RMSNorm-fold features
__device__ __forceinline__ void epilogue_w4(const GemmArgs& p, f32x4 (&acc)[2][8][4], int mrow0, int ncol0, int lane) {
  constexpr bool GEN = EV == 1, FOLD = EV == 2;
  int fr = lane & 15, g = lane >> 4;
  asm volatile("" : "+v"(fr), "+v"(g));      // opaque: keeps the lane offsets below from being hoisted out of the tile loop (and spilled across the K loop)
  auto pk = [&](int h, int i, int s2) __attribute__((always_inline)) -> u32x4 {
    // (explicit v_accvgpr_read per element: when the compiler copied the accumulators out of the AGPRs on its own it did so for all 256 at the
    // K loop's exit and spilled ~150 tuples to scratch)
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x, y;
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[h][i][2 * s2][e]));
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(acc[h][i][2 * s2 + 1][e]));
      a[e] = x; b[e] = y;
    }
    u32x4 v;
    v[0] = pack_bf16x2(a[0], a[1]); v[1] = pack_bf16x2(a[2], a[3]); v[2] = pack_bf16x2(b[0], b[1]); v[3] = pack_bf16x2(b[2], b[3]);
    return v;
  };
  // The epilogue is VALU-ISSUE bound (one wave per SIMD: every instruction costs its 4 cycles; a first version with 64-bit pointers and the
  // residual / cache-policy choices inside the store loop spent 25 instructions per 16-byte store: 4.1 us per tile even with only 32 blocks on
  // the chip).  Every matrix is therefore addressed through a buffer descriptor based at THIS WAVE's 128 x 128 sub-tile (scalar arithmetic):
  // lane offset (fr * ld + 8 g) elements, + 16 ld per A fragment, the column part in the instruction's immediate.
  auto tile_rs = [&](const void* base, long ld, long col) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
    const uint64_t b = (uint64_t)(uintptr_t)base + (uint64_t)(((long)mrow0 * ld + col) * 2);
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    // general variant: rows at and past M (the last row tile's tail) fall outside the descriptor: loads return 0, stores are dropped
    int nrec = 0x7fffffff;
    if constexpr (GEN) {
      const int rows_left = min(128, p.M - mrow0);
      nrec = __builtin_amdgcn_readfirstlane(rows_left > 0 ? (int)(((long)(rows_left - 1) * ld + 128) * 2) : 0);
    }
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)bu, 0, nrec, 0x00020000);
  };
  if constexpr (GEN) {
    // General epilogue (round 4): fp32 accumulator (* row scale) + bias -> bf16 -> activation -> bf16 -> + residual -> bf16, the rounding points of
    // store4 / HF's bf16 modules, on the whole-line store pattern of the lean path below.  Bias lives in 32 VGPRs per tile (the lane's 8 columns
    // of each (h, s2) block), the activation is a template parameter (no per-element switch), residual rows are prefetched like the lean path's.
    const bool hi8 = (fr & 8) != 0;
    const __amdgpu_buffer_rsrc_t crs = tile_rs(p.C, p.ldc, ncol0);
    const int cstep = p.ldc * 32;
    const int coff1 = ((fr & 7) * p.ldc + g * 8 + (hi8 ? 32 : 0)) * 2, coff2 = ((8 + (fr & 7)) * p.ldc + g * 8 + (hi8 ? 0 : 32)) * 2;
    if (p.rope_cos) {
      // QKV projection with RoPE in the epilogue (round 4): the wave's 128 columns are ONE head, and lane (fr, g) holds feature d = 32 s2 + 8 g + j
      // in its h = 0 registers and d + 64 in its h = 1 registers, so the rotation pairs never leave the lane.  Both halves of a 16-row block are
      // finished together (the loop over h is innermost here), rounding points as rope_kernel: bf(bf(x1 c) - bf(x2 s)), bf(bf(x2 c) + bf(x1 s)).
      const bool do_rope = ncol0 < p.rope_cols;
      u32x4 pz1 = u32x4{0u, 0u, 0u, 0u}, pz2 = pz1, pz3 = pz1, pz4 = pz1;      // the previous row block's store data (W4_KEEP2)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = mrow0 + 16 * i + fr;
        const int rowc = min(row, p.M - 1);
        const float rs = p.rowscale ? p.rowscale[rowc] : 1.f;
        const int pos = p.rope_pos ? p.rope_pos[rowc] : rowc % p.rope_S;
        u32x4 v[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float a0[8], a1[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0[e]) : "a"(acc[0][i][2 * s2][e]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0[4 + e]) : "a"(acc[0][i][2 * s2 + 1][e]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1[e]) : "a"(acc[1][i][2 * s2][e]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1[4 + e]) : "a"(acc[1][i][2 * s2 + 1][e]));
          }
          if (do_rope) {
            const float* cp = p.rope_cos + (long)pos * 64 + 32 * s2 + 8 * g;
            const float* sp = p.rope_sin + (long)pos * 64 + 32 * s2 + 8 * g;
            const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4), s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x1 = bfround(a0[e] * rs), x2 = bfround(a1[e] * rs);
              const float cc = e < 4 ? c0[e & 3] : c1[e & 3], ss = e < 4 ? s0[e & 3] : s1[e & 3];
              a0[e] = bfround(x1 * cc) + bfround(-x2 * ss);
              a1[e] = bfround(x2 * cc) + bfround(x1 * ss);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] *= rs; a1[e] *= rs; }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][s2][e] = pack_bf16x2(a0[2 * e], a0[2 * e + 1]);
            v[1][s2][e] = pack_bf16x2(a1[2 * e], a1[2 * e + 1]);
          }
        }
        W4_KEEP2(pz1, pz2); W4_KEEP2(pz3, pz4);          // (every accumulator read of this row block has issued)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 z1, z2;
          w4_rows8_swap(z1, z2, v[h][0], v[h][1]);
          const int so = __builtin_amdgcn_readfirstlane(i * cstep + h * 128);
          __builtin_amdgcn_raw_buffer_store_b128(z1, crs, coff1, so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(z2, crs, coff2, so, 0);
          if (h == 0) { pz1 = z1; pz2 = z2; } else { pz3 = z1; pz4 = z2; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    auto bodyg = [&](auto epi_c, auto res_c) __attribute__((always_inline)) {
      constexpr int EPI = decltype(epi_c)::value;
      constexpr bool RES = decltype(res_c)::value;
      float bf_[2][2][8];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u32x4 q = u32x4{0u, 0u, 0u, 0u};
          if (p.bias) q = *(const u32x4*)(p.bias + ncol0 + 64 * h + 32 * s2 + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bf_[h][s2][2 * e] = __builtin_bit_cast(float, q[e] << 16);
            bf_[h][s2][2 * e + 1] = __builtin_bit_cast(float, q[e] & 0xffff0000u);
          }
        }
      float rsv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = mrow0 + 16 * i + fr;
        rsv[i] = (p.rowscale && row < p.M) ? p.rowscale[row] : 1.f;
      }
      u32x4 rv[8][2];
      const __amdgpu_buffer_rsrc_t rrs = tile_rs(RES ? (const void*)p.res : p.C, RES ? p.ldr : p.ldc, ncol0);
      const int roff = (fr * p.ldr + g * 8) * 2, rstep = p.ldr * 32;
      if (RES) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) rv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + i * rstep, s2 * 64, 0);
      }
      u32x4 pz1 = u32x4{0u, 0u, 0u, 0u}, pz2 = pz1;       // the previous block's store data (W4_KEEP2)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          u32x4 xy[2];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            float a[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[e]) : "a"(acc[h][i][2 * s2][e]));
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[4 + e]) : "a"(acc[h][i][2 * s2 + 1][e]));
            }
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = a[e] * rsv[i] + bf_[h][s2][e];
              if (EPI != EPI_NONE) x = apply_epi_c<EPI>(bfround(x));
              a[e] = x;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(a[2 * e], a[2 * e + 1]);
            if (RES) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float lo = __builtin_bit_cast(float, v[e] << 16) + __builtin_bit_cast(float, rv[i][s2][e] << 16);
                const float hi = __builtin_bit_cast(float, v[e] & 0xffff0000u) + __builtin_bit_cast(float, rv[i][s2][e] & 0xffff0000u);
                v[e] = pack_bf16x2(lo, hi);
              }
              if (h == 0) rv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + i * rstep, 128 + s2 * 64, 0);
            }
            xy[s2] = v;
          }
          W4_KEEP2(pz1, pz2);
          u32x4 z1, z2;
          w4_rows8_swap(z1, z2, xy[0], xy[1]);
          const int so = __builtin_amdgcn_readfirstlane(i * cstep + h * 128);
          __builtin_amdgcn_raw_buffer_store_b128(z1, crs, coff1, so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(z2, crs, coff2, so, 0);
          pz1 = z1; pz2 = z2;
          __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    const int epi = p.epi & 0xff;
    if (p.res != nullptr) bodyg(std::integral_constant<int, EPI_NONE>{}, T_{});       // (activation + residual: the launcher keeps those on the 8-phase kernel)
    else if (epi == EPI_GELU) bodyg(std::integral_constant<int, EPI_GELU>{}, F_{});
    else if (epi == EPI_QUICK_GELU) bodyg(std::integral_constant<int, EPI_QUICK_GELU>{}, F_{});
    else if (epi == EPI_RELU) bodyg(std::integral_constant<int, EPI_RELU>{}, F_{});
    else bodyg(std::integral_constant<int, EPI_NONE>{}, F_{});
    return;
  }
#ifndef W4E_ONLY
#define W4E_ONLY 0
#endif
  if (!FOLD && (W4E_ONLY == 0 || W4E_ONLY == 3) && p.mode == 2) {
    // SwiGLU backward: the accumulators are d(act); act block (64 h + 32 s + 8 g ..+7) <-> the 32 bytes (g8 | u8) at twice that column of the saved
    // gate_up (aux) and of d(gate_up) (C).  The saved rows of half h = 1 are fetched into the registers half h = 0 has just consumed.
    const __amdgpu_buffer_rsrc_t ars = tile_rs(p.aux, p.ldaux, 2L * ncol0), drs = tile_rs(p.C, p.ldc, 2L * ncol0);
    const int aoff = (fr * p.ldaux + g * 16) * 2, astep = p.ldaux * 32, dstep = p.ldc * 32;
    const int doff1 = ((fr & 7) * p.ldc + g * 16 + ((fr & 8) ? 8 : 0)) * 2, doff2 = ((8 + (fr & 7)) * p.ldc + g * 16 + ((fr & 8) ? 0 : 8)) * 2;
    u32x4 gv[8][2], uv[8][2];
    u32x4 pz1 = u32x4{0u, 0u, 0u, 0u}, pz2 = pz1;         // the previous sub-block's store data (W4_KEEP2)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        gv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + i * astep, s2 * 128, 0);
        uv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + i * astep, s2 * 128 + 16, 0);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 dv = __builtin_bit_cast(bf16x8, pk(h, i, s2));
          W4_KEEP2(pz1, pz2);
          const bf16x8 gq = __builtin_bit_cast(bf16x8, gv[i][s2]), uq = __builtin_bit_cast(bf16x8, uv[i][s2]);
          bf16x8 og, ou;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gg = bf2f((bf16_t)gq[e]), uu = bf2f((bf16_t)uq[e]), dd = bf2f((bf16_t)dv[e]);
            const float sg = 1.f / (1.f + __expf(-gg));
            og[e] = (short)f2bf(dd * uu * sg * (1.f + gg * (1.f - sg)));
            ou[e] = (short)f2bf(dd * gg * sg);
          }
          // whole-line stores (see w4_rows8_swap): a row's 128-byte line of d(gate_up) is (og | ou) x 4 lanes g; rows 8..15 hand their ou up, take og's place
          u32x4 z1, z2;
          w4_rows8_swap(z1, z2, __builtin_bit_cast(u32x4, og), __builtin_bit_cast(u32x4, ou));
          const int so = __builtin_amdgcn_readfirstlane(i * dstep + h * 256 + s2 * 128);
          __builtin_amdgcn_raw_buffer_store_b128(z1, drs, doff1, so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(z2, drs, doff2, so, 0);
          pz1 = z1; pz2 = z2;
          if (h == 0) {
            gv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + i * astep, 256 + s2 * 128, 0);
            uv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + i * astep, 256 + s2 * 128 + 16, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    return;
  }
  const __amdgpu_buffer_rsrc_t crs = tile_rs(p.C, p.ldc, ncol0);
  const int cstep = p.ldc * 32;
  if ((W4E_ONLY == 0 || W4E_ONLY == 2) && p.mode == 1) {
    auto fwd_body = [&](auto rs_c) __attribute__((always_inline)) {
    constexpr bool RS = decltype(rs_c)::value;        // fp32 row scale on the accumulators (RMSNorm folded into the weight): gate_up = bf16(acc * scale)
    float rsv[8];
    if (RS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rsv[i] = p.rowscale[mrow0 + 16 * i + fr];
    }
    auto pks = [&](int h, int i, int s2) __attribute__((always_inline)) -> u32x4 {
      if (!RS) return pk(h, i, s2);
      f32x4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x, y;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[h][i][2 * s2][e]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(acc[h][i][2 * s2 + 1][e]));
        a[e] = x * rsv[i]; b[e] = y * rsv[i];
      }
      u32x4 v;
      v[0] = pack_bf16x2(a[0], a[1]); v[1] = pack_bf16x2(a[2], a[3]); v[2] = pack_bf16x2(b[0], b[1]); v[3] = pack_bf16x2(b[2], b[3]);
      return v;
    };
    // SwiGLU forward: C = gate_up (raw), C2 = act.  After the swap the even-g lanes hold pair g / 2, the odd-g lanes pair (g + 3) / 2 of the half.
    // Whole-line stores (see w4_rows8_swap): gate_up rows from the (x, y) halves of one (h, i); an act row's 128-byte line is its h = 0 and h = 1
    // halves, so both are computed before the pair of act stores.
    const int pq = (g & 1) ? (g + 3) >> 1 : g >> 1;
    const __amdgpu_buffer_rsrc_t ars = tile_rs(p.C2, p.ldc2, ncol0 >> 1);
    const bool hi8 = (fr & 8) != 0;
    const int coff1 = ((fr & 7) * p.ldc + g * 8 + (hi8 ? 32 : 0)) * 2, coff2 = ((8 + (fr & 7)) * p.ldc + g * 8 + (hi8 ? 0 : 32)) * 2;
    const int aoff1 = ((fr & 7) * p.ldc2 + pq * 8 + (hi8 ? 32 : 0)) * 2, aoff2 = ((8 + (fr & 7)) * p.ldc2 + pq * 8 + (hi8 ? 0 : 32)) * 2;
    const int astep = p.ldc2 * 32;
    u32x4 pz1 = u32x4{0u, 0u, 0u, 0u}, pz2 = pz1, pz3 = pz1, pz4 = pz1, pz5 = pz1, pz6 = pz1;      // the data of the last six stores (W4_KEEP2)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u32x4 oh[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 x = pks(h, i, 0), y = pks(h, i, 1);
        W4_KEEP2(pz1, pz2); W4_KEEP2(pz3, pz4); W4_KEEP2(pz5, pz6);
        u32x4 z1, z2;
        w4_rows8_swap(z1, z2, x, y);
        const int so = __builtin_amdgcn_readfirstlane(i * cstep + h * 128);
        __builtin_amdgcn_raw_buffer_store_b128(z1, crs, coff1, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(z2, crs, coff2, so, 0);
        if (h == 0) { pz1 = z1; pz2 = z2; } else { pz5 = z1; pz6 = z2; }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const auto r = __builtin_amdgcn_permlane16_swap(x[d], y[d], false, false);
          x[d] = r[0]; y[d] = r[1];
        }
        const bf16x8 gvv = __builtin_bit_cast(bf16x8, x), uvv = __builtin_bit_cast(bf16x8, y);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bfround(silu(bf2f((bf16_t)gvv[e]))) * bf2f((bf16_t)uvv[e]));
        oh[h] = __builtin_bit_cast(u32x4, o);
      }
      u32x4 a1, a2;
      w4_rows8_swap(a1, a2, oh[0], oh[1]);
      const int sa = __builtin_amdgcn_readfirstlane(i * astep);
      __builtin_amdgcn_raw_buffer_store_b128(a1, ars, aoff1, sa, 0);
      __builtin_amdgcn_raw_buffer_store_b128(a2, ars, aoff2, sa, 0);
      pz3 = a1; pz4 = a2;
      __builtin_amdgcn_sched_barrier(0);
    }
    };
    if constexpr (FOLD) fwd_body(std::true_type{}); else fwd_body(std::false_type{});
    return;
  }
  // plain / residual: the residual / non-temporal choices are made once per tile, not per store.
  // FULL-LINE stores (round 4, tools/probes/store_pattern_probe.hip): a buffer_store_dwordx4 whose lanes cover 16 rows x 64 B moves 15.8 B/clk
  // per CU, one that covers 8 rows x 128 B (whole cache lines) 52.9 B/clk.  The accumulator layout hands lane (fr, g) the 16 bytes at column
  // 8 g of row fr for the left (x: B fragments 0, 1) and the right (y: fragments 2, 3) 64-byte half of the row's line.  The lanes of rows 8..15
  // trade their x for the y of the lane 8 rows up (one DPP row_ror:8 inside each 16-lane row), so store 1 writes rows 0..7 and store 2 rows
  // 8..15 of the 16-row block as whole lines: lane (fr, g) -> store 1: row fr & 7, byte column (fr & 8 ? 64 : 0) + 16 g; store 2: row 8 + (fr & 7),
  // byte column (fr & 8 ? 0 : 64) + 16 g.  Pure data movement: results are bit-identical.
  if (W4E_ONLY > 1) return;
  const bool hi8 = (fr & 8) != 0;
  const int coff1 = ((fr & 7) * p.ldc + g * 8 + (hi8 ? 32 : 0)) * 2, coff2 = ((8 + (fr & 7)) * p.ldc + g * 8 + (hi8 ? 0 : 32)) * 2;
  auto body = [&](auto res_c, auto nt_c, auto ssq_c) __attribute__((always_inline)) {
    constexpr bool RES = decltype(res_c)::value, NT = decltype(nt_c)::value, SSQ = decltype(ssq_c)::value;
    // SSQ: one fp32 partial per (row, 64-column half h, lane g) = 16 columns, stored as soon as the (h, i) block is done (a value that lived across the
    // whole epilogue made the allocator move the accumulator chain out of the AGPRs: 150-190 spilled tuples), N / 16 partials per row, through a buffer
    // descriptor based at this wave's first row / first partial (scalar (i, h) offset: no 64-bit address arithmetic)
    const int npr = p.N >> 4;
    __amdgpu_buffer_rsrc_t srs = crs;
    int ssoff = 0;
    if (SSQ) {
      const uint64_t b = (uint64_t)(uintptr_t)p.sumsq_part + (uint64_t)(((long)mrow0 * npr + (ncol0 >> 4)) * 4);
      const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
      srs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)bu, 0, 0x7fffffff, 0x00020000);
      ssoff = (fr * npr + g) * 4;
    }
    u32x4 rv[8][2];
    u32x4 pz1 = u32x4{0u, 0u, 0u, 0u}, pz2 = pz1;         // the previous block's store data (W4_KEEP2)
    float psq = 0.f;                                      // ... and its sum-of-squares partial (a one-dword store of its own)
    const __amdgpu_buffer_rsrc_t rrs = tile_rs(RES ? (const void*)p.res : p.C, RES ? p.ldr : p.ldc, ncol0);
    const int roff = (fr * p.ldr + g * 8) * 2, rstep = p.ldr * 32;
    if (RES) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) rv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + i * rstep, s2 * 64, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u32x4 xy[2];
        float sq = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u32x4 v = pk(h, i, s2);
          if (RES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = __builtin_bit_cast(float, v[e] << 16) + __builtin_bit_cast(float, rv[i][s2][e] << 16);
              const float hi = __builtin_bit_cast(float, v[e] & 0xffff0000u) + __builtin_bit_cast(float, rv[i][s2][e] & 0xffff0000u);
              v[e] = pack_bf16x2(lo, hi);
              if (SSQ) sq = fmaf(lo, lo, fmaf(hi, hi, sq));
            }
            if (h == 0) rv[i][s2] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff + i * rstep, 128 + s2 * 64, 0);
          }
          xy[s2] = v;
        }
        W4_KEEP2(pz1, pz2);
        if (SSQ) W4_KEEP2(psq, psq);
        u32x4 z1, z2;
        w4_rows8_swap(z1, z2, xy[0], xy[1]);
        const int so = __builtin_amdgcn_readfirstlane(i * cstep + h * 128);      // row block + half in the scalar offset: no address VALU per store
        if (NT) {
          __builtin_amdgcn_raw_buffer_store_b128(z1, crs, coff1, so, 2);
          __builtin_amdgcn_raw_buffer_store_b128(z2, crs, coff2, so, 2);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(z1, crs, coff1, so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(z2, crs, coff2, so, 0);
        }
        if (SSQ) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, sq), srs, ssoff, __builtin_amdgcn_readfirstlane((i * 16 * npr + h * 4) * 4), 0);
        pz1 = z1; pz2 = z2; psq = sq;
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  using T_ = std::true_type; using F_ = std::false_type;
  if constexpr (FOLD) {
    if (p.c_nt) body(T_{}, T_{}, T_{}); else body(T_{}, F_{}, T_{});      // (the launcher sends residual + sumsq launches only)
  } else {
    if (p.res != nullptr) { if (p.c_nt) body(T_{}, T_{}, F_{}); else body(T_{}, F_{}, F_{}); }
    else { if (p.c_nt) body(F_{}, T_{}, F_{}); else body(F_{}, F_{}, F_{}); }
  }
}

template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_nt_256(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;                    // [buf 0: A 256x64 | B 256x64][buf 1: ...]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
  const int ntiles = tiles_m * tiles_n;
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, g = lane >> 4;
  const int nt = p.K >> 6;

  // per-thread staging geometry (same for every tile): 4 chunks per operand, LDS chunk q = (row q>>3, slot q&7)
  int rowi[4], gci[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = it * 512 + tid;
    rowi[it] = q >> 3;
    gci[it] = ((q & 7) ^ ((rowi[it] >> 1) & 7)) * 8;
  }
  const bf16_t* curA[4];     // this thread's 4 + 4 source rows of the output tile whose K-tiles are being streamed
  const bf16_t* curB[4];
#define SET_TILE(TC)                                                               \
  _Pragma("unroll") for (int it = 0; it < 4; ++it) {                               \
    curA[it] = p.A + (long)min((TC).m0 + rowi[it], p.M - 1) * p.lda + gci[it];     \
    curB[it] = p.B + (long)min((TC).n0 + rowi[it], p.N - 1) * p.ldb + gci[it];     \
  }
  int v = blockIdx.x;
  TileCoord tc = tile_coord_256(v, tiles_m, tiles_n);
  SET_TILE(tc);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    GLDS16(curA[it], smem + (it * 512 + wave * 64) * 8);
    GLDS16(curB[it], smem + 16384 + (it * 512 + wave * 64) * 8);
  }
  __syncthreads();
  int cur = 0;
  while (true) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int vnext = v + gridDim.x;
    const TileCoord tcur = tc;
    for (int t = 0; t < nt; ++t) {
      // source of the NEXT K-tile in the stream: same output tile, or the first K-tile of this block's next output tile
      // (the very last one re-fetches a valid dummy so the DMA pattern stays branch-free)
      long koff = (long)(t + 1) * 64;
      if (t + 1 == nt) {
        koff = 0;
        if (vnext < ntiles) {
          tc = tile_coord_256(vnext, tiles_m, tiles_n);
          SET_TILE(tc);
        }
      }
      const bf16_t* As = smem + cur * 32768;
      const bf16_t* Bs = As + 16384;
      bf16_t* Asn = smem + (cur ^ 1) * 32768;
      bf16_t* Bsn = Asn + 16384;
      // 64 MFMAs per K-tile; the 8 LDS-DMA instructions of the next K-tile are spread one per 4 MFMAs over the first half: a global_load_lds
      // blocks its wave for ~60-100 issue cycles, which the SIMD's other wave covers only if the stalls are not bunched up.
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xf[8], wf[4];
        const int cg = ks * 4 + g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = wc * 64 + j * 16 + fr;
          wf[j] = *(const bf16x8*)(Bs + r * 64 + ((cg ^ ((r >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = wr * 128 + i * 16 + fr;
          xf[i] = *(const bf16x8*)(As + r * 64 + ((cg ^ ((r >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
          if (ks == 0) {                                       // all 8 DMAs in the FIRST half: the second half (32 MFMAs) is landing time
            const int it = (i >> 1);                           // 0..3
            if (i & 1) GLDS16(curB[it] + koff, Bsn + (it * 512 + wave * 64) * 8);
            else GLDS16(curA[it] + koff, Asn + (it * 512 + wave * 64) * 8);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ... then 1 VMEM read (the LDS-DMA)
          }
        }
      }
      __syncthreads();
      cur ^= 1;
    }
    // buf[cur] now holds (or is receiving) the next tile's first K-tile; buf[cur ^ 1] was just consumed -> C staging
    if (!OUT_F32) {
      epilogue_swz<2>(p, smem + (cur ^ 1) * 32768 + wave * 4096, acc, tcur.m0 + wr * 128, tcur.n0 + wc * 64, lane);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          store4<OUT_F32>(p, tcur.m0 + wr * 128 + i * 16 + fr, tcur.n0 + wc * 64 + j * 16 + g * 4, acc[i][j]);
    }
    if (vnext >= ntiles) break;
    v = vnext;
    __syncthreads();             // every wave is done with the staging slice before the next DMA may land in it
  }
#undef SET_TILE
}

// raw workgroup barrier fenced against the instruction scheduler (used by the 8-phase kernel below)
#define VP_SB() __builtin_amdgcn_sched_barrier(0)
#define VP_BAR()                      \
  do {                                \
    VP_SB();                          \
    __builtin_amdgcn_s_barrier();     \
    VP_SB();                          \
  } while (0)

// ------------------------------------------------------------------------------------------------
// 8-phase ping-pong kernel (two K-tiles = 8 phases per loop trip).  Same 256x256x64 tile / 8 waves / 2 LDS buffers as
// above; what changes is the *granularity and balance* of the pipeline:
//   * a phase = [R: <= 12 ds_read_b128 + 2 global_load_lds] barrier [M: 16 MFMAs on one 64x32 quadrant] barrier, and
//     the second wave group runs one barrier behind, so every barrier interval pairs one group's R with the other's M
//     and both are ~300 cycles long (the coarse ping-pong above had an 1100-cycle R0 against a 512-cycle M);
//   * the next K-tile is DMA'd as four 16 KB *pieces* ordered by first use — A rows of the waves' first 64-row half,
//     B rows of their first 32-column half, the other B half, the other A half — one piece per phase, so every piece
//     has >= 3 phases to land and the queue is never drained: s_waitcnt vmcnt(4) keeps two pieces in flight across
//     every barrier (the wait at the end of phase k's R retires exactly the piece phase k+1 reads first).
// ------------------------------------------------------------------------------------------------
#define STAMP(I) if ((p.dbg & 0x10000) && threadIdx.x == 0 && first_tile) vp_dbg_stamps[blockIdx.x * 8 + (I)] = wall_clock64();
template <bool OUT_F32, bool PH4 = false>      // PH4: 4 phases per K-tile (32 MFMAs each, half the barriers), see the loop
__global__ __launch_bounds__(512) void gemm_nt_256p8(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;                    // [buf][A 256x64 | B 256x64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
  const int ntiles = tiles_m * tiles_n;
  const int wr = __builtin_amdgcn_readfirstlane(wave >> 2), wc = wave & 3;
  const int fr = lane & 15, g = lane >> 4;
  const int nt = p.K >> 6;                             // even when the grid is persistent (checked by the launcher)

  // staging pieces (128 rows x 64 k = 16 KB = 2 chunks per thread each):
  //   piece 0: A rows wr'*128 + [0,64)      piece 1: B rows wc'*64 + [0,32)
  //   piece 2: B rows wc'*64 + [32,64)      piece 3: A rows wr'*128 + [64,128)
  int ldsoff[4][2];                                    // wave-uniform LDS offsets (SGPRs)
#pragma unroll
  for (int pc = 0; pc < 4; ++pc)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pr0 = (it * 512 + wave * 64) >> 3;     // first of the 8 rows this instruction covers
      int row0;
      if (pc == 0) row0 = (pr0 >> 6) * 128 + (pr0 & 63);
      else if (pc == 3) row0 = (pr0 >> 6) * 128 + 64 + (pr0 & 63);
      else if (pc == 1) row0 = (pr0 >> 5) * 64 + (pr0 & 31);
      else row0 = (pr0 >> 5) * 64 + 32 + (pr0 & 31);
      ldsoff[pc][it] = ((pc == 0 || pc == 3) ? 0 : 16384) + row0 * 64;
    }
  const bf16_t* src[4][2];                             // source of the NEXT K-tile in the stream (per piece); += 64 per K-tile
  // (re)computed from the thread id at every output-tile switch; the asm keeps the index math from being hoisted and held
  // in registers across the K loop
#define SET_SRC(TC, K0)                                                            \
  {                                                                                \
    int tid_ = tid;                                                                \
    asm volatile("" : "+v"(tid_));                                                 \
    _Pragma("unroll") for (int pc = 0; pc < 4; ++pc)                               \
      _Pragma("unroll") for (int it = 0; it < 2; ++it) {                           \
        const int q = it * 512 + tid_;                                             \
        const int pr = q >> 3;                                                     \
        int row;                                                                   \
        if (pc == 0) row = (pr >> 6) * 128 + (pr & 63);                            \
        else if (pc == 3) row = (pr >> 6) * 128 + 64 + (pr & 63);                  \
        else if (pc == 1) row = (pr >> 5) * 64 + (pr & 31);                        \
        else row = (pr >> 5) * 64 + 32 + (pr & 31);                                \
        const int gc = ((q & 7) ^ ((row >> 1) & 7)) * 8 + (K0) * 64;               \
        src[pc][it] = (pc == 0 || pc == 3)                                         \
            ? p.A + (long)min((TC).m0 + row, p.M - 1) * p.lda + gc                 \
            : p.B + (long)min((TC).n0 + row, p.N - 1) * p.ldb + gc;                \
      }                                                                            \
  }
#define ISSUE_PIECE(PC, BUF)                                                       \
  {                                                                                \
    bf16_t* base_ = smem + (BUF) * 32768;                                          \
    GLDS16(src[PC][0], base_ + ldsoff[PC][0]);                                     \
    GLDS16(src[PC][1], base_ + ldsoff[PC][1]);                                     \
  }
#define RD_A(DST, MH)                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
      const int r = wr * 128 + (MH) * 64 + i * 16 + fr;                            \
      DST[ks][i] = *(const bf16x8*)(As + r * 64 + (((ks * 4 + g) ^ ((r >> 1) & 7)) << 3)); \
    }
#define RD_B(DST, NH)                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                \
      const int r = wc * 64 + (NH) * 32 + j * 16 + fr;                             \
      DST[ks][j] = *(const bf16x8*)(Bs + r * 64 + (((ks * 4 + g) ^ ((r >> 1) & 7)) << 3)); \
    }
#define MM(XA, WB, MH, NH)                                                         \
  {                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                              \
          acc[(MH) * 4 + i][(NH) * 2 + j] =                                        \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB[ks][j], XA[ks][i], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                 \
  }
#define END_R() asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); VP_BAR();     /* ds_reads keep flying across the barrier; the MFMAs wait for them */

  int v = (int)blockIdx.x;                               // current output tile
  bool first_tile = true;
  __shared__ int s_next;
  const bool dyn = p.sched != nullptr && gridDim.x == 256;
  STAMP(0);
  int sbm = 0, sbn = 0;                                // super-block walk when the (persistent) grid and the tile grid allow it
  if (gridDim.x == 256 && !(p.dbg & 0x40000)) {
    if (tiles_m % 16 == 0 && tiles_n % 16 == 0) { sbm = 16; sbn = 16; }
    else if (tiles_m % 32 == 0 && tiles_n % 8 == 0) { sbm = 32; sbn = 8; }
  }
#define TILE_OF(V) (sbm ? tile_coord_sb((V), tiles_m, tiles_n, sbm, sbn) : tile_coord_256((V), tiles_m, tiles_n))
  TileCoord tc = TILE_OF(v);
  SET_SRC(tc, 0);
  ISSUE_PIECE(0, 0);
  ISSUE_PIECE(1, 0);
  ISSUE_PIECE(2, 0);
  ISSUE_PIECE(3, 0);
  if (PH4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // pieces 0,1 of K-tile 0 landed (this wave's parts)
  VP_BAR();
  bf16x8 xa[2][4], wb0[2][2], wb1[2][2], xn[4];
  if (!PH4) {                                          // A(m0, ks=0) fragments of K-tile 0 (later ones are prefetched in phase 4)
    const bf16_t* As = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wr * 128 + i * 16 + fr;
      xn[i] = *(const bf16x8*)(As + r * 64 + ((g ^ ((r >> 1) & 7)) << 3));
    }
  }
  while (true) {
    STAMP(1);
    if ((p.dbg & 0x10000) && threadIdx.x == 0 && first_tile) vp_dbg_stamps[blockIdx.x * 8 + 6] = clock64();   // shader cycles
    f32x4 acc[8][4];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wr == 1) VP_BAR();                             // stagger: group 1 runs one barrier behind
    const TileCoord tcur = tc;
    int vnext = v + (int)gridDim.x;
    // Dynamic scheduling (used when another kernel, e.g. an RCCL collective, may hold some CUs: a block that starts late would otherwise
    // do its whole static share after everyone else has finished).  Each XCD's 32 resident blocks claim the XCD's tiles in order from a
    // per-XCD counter, so tile v still runs on XCD v & 7 and the super-block walk keeps its L2 residency.  The claim for the NEXT tile is made
    // here, at the start of the current one; every wave reads it a K loop (many barriers) later.
    if (dyn && tid == 0) {
      const int kq = 32 + atomicAdd(p.sched + (blockIdx.x & 7), 1);      // (counters rest at 0: the first 32 tiles of an XCD are the blocks' static ones)
      s_next = (kq >> 5) * 256 + (kq & 31) * 8 + (int)(blockIdx.x & 7);
    }
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      const bf16_t* As = smem + cur * 32768;
      const bf16_t* Bs = As + 16384;
      // the K-tile stream continues into this block's NEXT output tile (or a harmless re-fetch at the very end)
      if (t + 1 < nt) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) { src[pc][0] += 64; src[pc][1] += 64; }
      } else {
        if (dyn) vnext = __builtin_amdgcn_readfirstlane(*(volatile int*)&s_next);
        if (vnext < ntiles) tc = TILE_OF(vnext);
        SET_SRC(tc, 0);
      }
      if (PH4) {
        // 4-phase variant: a piece issued in the R part of phase X is retired for BOTH wave groups only after the barrier that ends the M part
        // of phase X+1 (group 1 runs one barrier behind), so it may be read from the R part of phase X+2 = the same phase of the next K-tile.
        // ---- phase A: quadrants (m0, n0), (m0, n1): 32 MFMAs.  issues pieces 0,1,2 of the next K-tile; reads A(m0) x8, B(n0) x4, B(n1) x4
        ISSUE_PIECE(0, cur ^ 1);
        ISSUE_PIECE(1, cur ^ 1);
        ISSUE_PIECE(2, cur ^ 1);
        RD_A(xa, 0);
        RD_B(wb0, 0);
        RD_B(wb1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the previous phase B's piece 3 of THIS K-tile landed (this wave's part)
        VP_BAR();
        MM(xa, wb0, 0, 0);
        MM(xa, wb1, 0, 1);
        VP_BAR();
        // ---- phase B: quadrants (m1, n1), (m1, n0): 32 MFMAs.  issues piece 3; reads A(m1) x8
        ISSUE_PIECE(3, cur ^ 1);
        RD_A(xa, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // phase A's pieces 0,1,2 of the next K-tile landed
        VP_BAR();
        MM(xa, wb1, 1, 1);
        MM(xa, wb0, 1, 0);
        VP_BAR();
        continue;
      }
      // ---- phase 1: quadrant (m0, n0).  reads: B(n0) x4, A(m0, ks=1) x4  (A(m0, ks=0) came from the previous phase 4)
      ISSUE_PIECE(0, cur ^ 1);
      RD_B(wb0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wr * 128 + i * 16 + fr;
        xa[0][i] = xn[i];
        xa[1][i] = *(const bf16x8*)(As + r * 64 + (((4 + g) ^ ((r >> 1) & 7)) << 3));
      }
      END_R();
      MM(xa, wb0, 0, 0);
      VP_BAR();
      // ---- phase 2: quadrant (m0, n1).  reads: B(n1) x4
      ISSUE_PIECE(1, cur ^ 1);
      RD_B(wb1, 1);
      END_R();
      MM(xa, wb1, 0, 1);
      VP_BAR();
      // ---- phase 3: quadrant (m1, n1).  reads: A(m1) x8
      ISSUE_PIECE(2, cur ^ 1);
      RD_A(xa, 1);
      END_R();
      MM(xa, wb1, 1, 1);
      VP_BAR();
      // ---- phase 4: quadrant (m1, n0) (B(n0) still in registers).  reads: NEXT K-tile's A(m0, ks=0) x4 — its piece was
      // issued in phase 1 and retired for every wave by the vmcnt(4) at the end of phase 3's R section.
      ISSUE_PIECE(3, cur ^ 1);
      {
        const bf16_t* Asn = smem + (cur ^ 1) * 32768;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = wr * 128 + i * 16 + fr;
          xn[i] = *(const bf16x8*)(Asn + r * 64 + ((g ^ ((r >> 1) & 7)) << 3));
        }
      }
      END_R();
      MM(xa, wb0, 1, 0);
      VP_BAR();
    }
    if (wr == 0) VP_BAR();                             // re-align the two groups at the output-tile boundary
    STAMP(2);
    if ((p.dbg & 0x10000) && threadIdx.x == 0 && first_tile) vp_dbg_stamps[blockIdx.x * 8 + 7] = clock64();
    // buffer 1 (the last K-tile's, nt is even) is free for C staging; buffer 0 is receiving the next tile's first K-tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!OUT_F32) {
      epilogue_swz<2>(p, smem + ((nt - 1) & 1) * 32768 + wave * 4096, acc, tcur.m0 + wr * 128, tcur.n0 + wc * 64, lane);
    } else {
      // fp32 output (weight gradients): interior, plain tiles go straight from the accumulators as 16-byte stores
      const int mr = tcur.m0 + wr * 128, nc = tcur.n0 + wc * 64;
      const bool lean = __builtin_amdgcn_readfirstlane((int)(!p.bias && !p.res && (p.epi & 0xff) == EPI_NONE && mr + 128 <= p.M &&
                                                             nc + 64 <= p.N && (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0));
      if (lean) {
        float* c = (float*)p.C + (long)(mr + fr) * p.ldc + nc + g * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) *(f32x4*)(c + (long)(i * 16) * p.ldc + j * 16) = acc[i][j];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) store4<OUT_F32>(p, mr + i * 16 + fr, nc + j * 16 + g * 4, acc[i][j]);
      }
    }
    STAMP(3);
    if (p.dbg & 0x10000) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(4); }
    first_tile = false;
    if (vnext >= ntiles) break;
    v = vnext;
    VP_BAR();                                          // every wave is done with the staging slices before the next DMA lands there
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the final dummy DMAs must not outlive the workgroup's LDS
  if (dyn && tid == 0) {                               // the last block out re-arms the counters for the next launch on this stream
    __threadfence();
    if (atomicAdd(p.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 8; ++x) p.sched[x] = 0;
      p.sched[8] = 0;
      __threadfence();
    }
  }
  first_tile = true;
  STAMP(5);
#undef SET_SRC
#undef TILE_OF
#undef ISSUE_PIECE
#undef RD_A
#undef RD_B
#undef MM
#undef END_R
}

// Two ds_read_b64_tr_b16 (byte offsets OFF and OFF + GAP from a per-lane LDS byte address): this lane's 8 contraction slots (slot j <->
// tile row 16*(j>>2) + 4g + (j&3), the same mapping for both operands) of output feature (lane & 15) of the 16-feature block addressed.
// Inline asm on purpose: through the builtin the compiler assumes the read may alias the in-flight LDS-DMA writes and puts
// `s_waitcnt vmcnt(0)` in front of every batch (measured: 480 instead of 1150 TFLOP/s).  The caller waits (TR_WAIT) before use.
template <int OFF, int GAP>
static __device__ __forceinline__ bf16x8 trfrag2(uint32_t addr) {
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + GAP));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
static __device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// ------------------------------------------------------------------------------------------------
// TN variant of the 8-phase kernel: C[M,N] = A[K,M]^T B[K,N] with both operands contraction-major (weight gradients dW = dY^T X straight
// from the activations).  Same persistent / ping-pong / piece-ordered DMA structure; only the LDS image and the fragment reads differ.
// ------------------------------------------------------------------------------------------------
template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_tn_256p8(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;                    // [buf][A 256x64 | B 256x64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
  const int ntiles = tiles_m * tiles_n;
  const int wr = __builtin_amdgcn_readfirstlane(wave >> 2), wc = wave & 3;
  const int fr = lane & 15, g = lane >> 4;
  const int nt = p.K >> 6;                             // even when the grid is persistent (checked by the launcher)
  const int trow = 4 * g + (fr >> 2), tb4 = (lane & 3) * 4;          // transposing read: this lane's row / 8-byte piece
  const int swA = ((g & 1) << 1) | (fr >> 3), swB = g & 1;           // swizzle phases of that row (k>>1)&3, (k>>2)&1

  // TN operands: A is [K, M], B is [K, N] (the contraction index runs over ROWS: wgrad = dY^T X without transposing anything).
  // LDS per buffer: A part [m-half MH][wr'][64 k][64 m] (128-byte rows), B part at +16384: [n-half NH][wc'][64 k][32 n] (64-byte rows);
  // pieces in consumption order as in the NT kernel: 0 = A(MH 0), 1 = B(NH 0), 2 = B(NH 1), 3 = A(MH 1); 16 KB = 2 chunks per thread each.
  // Fragments contract over tile ROWS, so they come from the transposing ds_read_b64_tr_b16; the 32-byte unit index of a row is XOR-ed
  // with (k>>1)&3 (A) / (k>>2)&1 (B) on the SOURCE side so that the 16 rows of one read spread over all banks (2-way = optimal).
  int ldsoff[4][2];                                    // wave-uniform LDS offsets (SGPRs)
#pragma unroll
  for (int pc = 0; pc < 4; ++pc)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q0 = it * 512 + wave * 64;
      if (pc == 0 || pc == 3) ldsoff[pc][it] = (((pc == 3) * 2 + (q0 >> 9)) * 64 + ((q0 >> 3) & 63)) * 64;
      else ldsoff[pc][it] = 16384 + (((pc == 2) * 4 + (q0 >> 8)) * 64 + ((q0 >> 2) & 63)) * 32;
    }
  const bf16_t* src[4][2];                             // source of the NEXT K-tile in the stream (per piece); += 64 rows per K-tile
  const long stepA = 64 * p.lda, stepB = 64 * p.ldb;
#define SET_SRC(TC)                                                                \
  {                                                                                \
    int tid_ = tid;                                                                \
    asm volatile("" : "+v"(tid_));                                                 \
    _Pragma("unroll") for (int pc = 0; pc < 4; ++pc)                               \
      _Pragma("unroll") for (int it = 0; it < 2; ++it) {                           \
        const int q = it * 512 + tid_;                                             \
        if (pc == 0 || pc == 3) {                                                  \
          const int k = (q >> 3) & 63, cc = q & 7;                                 \
          const int col = (q >> 9) * 128 + (pc == 3) * 64 + (((cc >> 1) ^ ((k >> 1) & 3)) << 4) + (cc & 1) * 8; \
          src[pc][it] = p.A + (long)k * p.lda + (TC).m0 + col;                     \
        } else {                                                                   \
          const int k = (q >> 2) & 63, cc = q & 3;                                 \
          const int col = (q >> 8) * 64 + (pc == 2) * 32 + (((cc >> 1) ^ ((k >> 2) & 1)) << 4) + (cc & 1) * 8; \
          src[pc][it] = p.B + (long)k * p.ldb + (TC).n0 + col;                     \
        }                                                                          \
      }                                                                            \
  }
#define ISSUE_PIECE(PC, BUF)                                                       \
  {                                                                                \
    bf16_t* base_ = smem + (BUF) * 32768;                                          \
    GLDS16(src[PC][0], base_ + ldsoff[PC][0]);                                     \
    GLDS16(src[PC][1], base_ + ldsoff[PC][1]);                                     \
  }
  // per-lane LDS byte addresses of the fragment reads in buffer 0 (everything else is an instruction immediate, + buffer << 16)
  uint32_t abase[4], bbase[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) abase[i] = lds_addr(smem) + 2 * ((wr * 64 + trow) * 64 + ((i ^ swA) << 4) + tb4);
#pragma unroll
  for (int j = 0; j < 2; ++j) bbase[j] = lds_addr(smem) + 2 * (16384 + (wc * 64 + trow) * 32 + ((j ^ swB) << 4) + tb4);
#define RD_A(DST, MH)                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                  \
    DST[0][i] = trfrag2<(MH) * 16384, 2048>(abase[i] + co);                        \
    DST[1][i] = trfrag2<(MH) * 16384 + 4096, 2048>(abase[i] + co);                 \
  }
#define RD_B(DST, NH)                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                  \
    DST[0][j] = trfrag2<(NH) * 16384, 1024>(bbase[j] + co);                        \
    DST[1][j] = trfrag2<(NH) * 16384 + 2048, 1024>(bbase[j] + co);                 \
  }
#define PIN(F) asm volatile("" : "+v"(F))
#define TR_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define MM(XA, WB, MH, NH)                                                         \
  {                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                              \
          acc[(MH) * 4 + i][(NH) * 2 + j] =                                        \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB[ks][j], XA[ks][i], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                 \
  }
#define END_R() asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); VP_BAR();     /* ds_reads keep flying across the barrier; the MFMAs wait for them */

  int v = blockIdx.x;
  __shared__ int s_next;
  const bool dyn = p.sched != nullptr && gridDim.x == 256;
  int sbm = 0, sbn = 0;                                // super-block walk when the (persistent) grid and the tile grid allow it
  if (gridDim.x == 256 && !(p.dbg & 0x40000)) {
    if (tiles_m % 16 == 0 && tiles_n % 16 == 0) { sbm = 16; sbn = 16; }
    else if (tiles_m % 32 == 0 && tiles_n % 8 == 0) { sbm = 32; sbn = 8; }
  }
#define TILE_OF(V) (sbm ? tile_coord_sb((V), tiles_m, tiles_n, sbm, sbn) : tile_coord_256((V), tiles_m, tiles_n))
  TileCoord tc = TILE_OF(v);
  SET_SRC(tc);
  ISSUE_PIECE(0, 0);
  ISSUE_PIECE(1, 0);
  ISSUE_PIECE(2, 0);
  ISSUE_PIECE(3, 0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // pieces 0,1 of K-tile 0 landed (this wave's parts)
  VP_BAR();
  bf16x8 xa[2][4], wb0[2][2], wb1[2][2], xn[4];
  {                                                    // A(m0, ks=0) fragments of K-tile 0 (later ones are prefetched in phase 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[i] = trfrag2<0, 2048>(abase[i]);
    TR_WAIT();
#pragma unroll
    for (int i = 0; i < 4; ++i) PIN(xn[i]);
  }
  while (true) {
    if (wr == 1) VP_BAR();                             // stagger: group 1 runs one barrier behind
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const TileCoord tcur = tc;
    int vnext = v + gridDim.x;
    if (dyn && tid == 0) {                               // dynamic per-XCD tile claim for the NEXT tile (see gemm_nt_256p8)
      const int kq = 32 + atomicAdd(p.sched + (blockIdx.x & 7), 1);      // (counters rest at 0: the first 32 tiles of an XCD are the blocks' static ones)
      s_next = (kq >> 5) * 256 + (kq & 31) * 8 + (int)(blockIdx.x & 7);
    }
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      const uint32_t co = (uint32_t)cur << 16;        // byte offset of the current buffer
      // the K-tile stream continues into this block's NEXT output tile (or a harmless re-fetch at the very end)
      if (t + 1 < nt) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) { const long st_ = (pc == 0 || pc == 3) ? stepA : stepB; src[pc][0] += st_; src[pc][1] += st_; }
      } else {
        if (dyn) vnext = __builtin_amdgcn_readfirstlane(*(volatile int*)&s_next);
        if (vnext < ntiles) tc = TILE_OF(vnext);
        SET_SRC(tc);
      }
      // ---- phase 1: quadrant (m0, n0).  reads: B(n0) x4, A(m0, ks=1) x4  (A(m0, ks=0) came from the previous phase 4)
      ISSUE_PIECE(0, cur ^ 1);
      RD_B(wb0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[0][i] = xn[i];
        xa[1][i] = trfrag2<4096, 2048>(abase[i] + co);
      }
      END_R();
      TR_WAIT();
#pragma unroll
      for (int q = 0; q < 2; ++q) { PIN(wb0[0][q]); PIN(wb0[1][q]); }
#pragma unroll
      for (int i = 0; i < 4; ++i) PIN(xa[1][i]);
      MM(xa, wb0, 0, 0);
      VP_BAR();
      // ---- phase 2: quadrant (m0, n1).  reads: B(n1) x4
      ISSUE_PIECE(1, cur ^ 1);
      RD_B(wb1, 1);
      END_R();
      TR_WAIT();
#pragma unroll
      for (int q = 0; q < 2; ++q) { PIN(wb1[0][q]); PIN(wb1[1][q]); }
      MM(xa, wb1, 0, 1);
      VP_BAR();
      // ---- phase 3: quadrant (m1, n1).  reads: A(m1) x8
      ISSUE_PIECE(2, cur ^ 1);
      RD_A(xa, 1);
      END_R();
      TR_WAIT();
#pragma unroll
      for (int i = 0; i < 4; ++i) { PIN(xa[0][i]); PIN(xa[1][i]); }
      MM(xa, wb1, 1, 1);
      VP_BAR();
      // ---- phase 4: quadrant (m1, n0) (B(n0) still in registers).  reads: NEXT K-tile's A(m0, ks=0) x4 — its piece was
      // issued in phase 1 and retired for every wave by the vmcnt(4) at the end of phase 3's R section.
      ISSUE_PIECE(3, cur ^ 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = trfrag2<0, 2048>(abase[i] + (co ^ 65536u));
      END_R();
      TR_WAIT();
#pragma unroll
      for (int i = 0; i < 4; ++i) PIN(xn[i]);
      MM(xa, wb0, 1, 0);
      VP_BAR();
    }
    if (wr == 0) VP_BAR();                             // re-align the two groups at the output-tile boundary
    // buffer 1 (the last K-tile's, nt is even) is free for C staging; buffer 0 is receiving the next tile's first K-tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!OUT_F32) {
      epilogue_swz<2>(p, smem + ((nt - 1) & 1) * 32768 + wave * 4096, acc, tcur.m0 + wr * 128, tcur.n0 + wc * 64, lane);
    } else {
      // fp32 output (weight gradients): interior, plain tiles go straight from the accumulators as 16-byte stores
      const int mr = tcur.m0 + wr * 128, nc = tcur.n0 + wc * 64;
      const bool lean = __builtin_amdgcn_readfirstlane((int)(!p.bias && !p.res && (p.epi & 0xff) == EPI_NONE && mr + 128 <= p.M &&
                                                             nc + 64 <= p.N && (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0));
      if (lean) {
        float* c = (float*)p.C + (long)(mr + fr) * p.ldc + nc + g * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4* dst = (f32x4*)(c + (long)(i * 16) * p.ldc + j * 16);
            *dst = (p.mode == 3) ? acc[i][j] + *dst : acc[i][j];       // mode 3: C += (gradient accumulation over token chunks)
          }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) store4<OUT_F32>(p, mr + i * 16 + fr, nc + j * 16 + g * 4, acc[i][j]);
      }
    }
    if (vnext >= ntiles) break;
    v = vnext;
    VP_BAR();                                          // every wave is done with the staging slices before the next DMA lands there
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the final dummy DMAs must not outlive the workgroup's LDS
  if (dyn && tid == 0) {                               // the last block out re-arms the counters for the next launch on this stream
    __threadfence();
    if (atomicAdd(p.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 8; ++x) p.sched[x] = 0;
      p.sched[8] = 0;
      __threadfence();
    }
  }
#undef SET_SRC
#undef TILE_OF
#undef ISSUE_PIECE
#undef PIN
#undef TR_WAIT
#undef RD_A
#undef RD_B
#undef MM
#undef END_R
}

// ------------------------------------------------------------------------------------------------
// 4-wave kernel (force code 8 / VP_GEMM_W4): ONE wave per SIMD, each owning a 128x128 quarter of the 256x256 output tile with its 256
// accumulator registers in AGPRs, every LDS fragment feeding 8 MFMAs (128 KB instead of 192 KB of LDS reads per K-tile and CU).
//
// Round 3 rewrite of the K loop.  What the rocprofv3 counters say (profiles/r03_gemm_vs_hipblaslt_pmc.txt; GRBM_GUI_ACTIVE / kernel time =
// the clock a kernel sustains): hipBLASLt's kernels for the decoder shapes win on CYCLES, not on clock — 2300-2520 GPU cycles per K-tile and
// CU at 1.72-1.84 GHz against 2810-3030 cycles at 1.89-1.99 GHz for the 8-phase kernel, whose eight waves spend 31 % of their cycles parked
// at its barriers (SQ_WAIT_ANY; theirs: 6 %).  A lone wave per SIMD never waits for a partner; what it must not do is put more than ONE
// non-MFMA instruction between two MFMAs (a 16x16x32 MFMA leaves ~3 issue slots): round 2's version of this kernel issued its 16 LDS-DMA
// pieces in bursts of [buffer_load, s_mov m0, s_add, s_add] inside one gap, all of them in the second half of the K-tile, behind a
// vmcnt(0), and paid ~30 cycles per piece (168 k cycles per 64 K-tiles against 137 k without the DMA).  The loop below is scheduled by hand:
//   * every gap between two MFMAs holds at most one other instruction: a ds_read_b128, an LDS-DMA piece, the m0 bump of the previous piece,
//     a wait or a barrier — and nothing the compiler generates (the per-piece global offsets live in 16 loop-invariant VGPRs, the K advance
//     in ONE SGPR bumped once per K-tile, the LDS destination in m0 bumped by asm);
//   * the two LDS buffers release their A and B halves separately (the fragments of K-step 1 are in registers long before the K-tile's
//     MFMAs are done), so the DMA of K-tile kt+2 starts a quarter into K-tile kt and is spread over ~50 gaps;
//   * vmcnt is only ever counted (16 pieces stay in flight across the hand-over barrier), never drained.
// Three barriers per K-tile (A half free | B half free | K-tile kt+1 landed), all four waves run the same stream, so they arrive together.
// The K-tile stream runs on into the block's next output tile; C staging has its own 32 KB of LDS behind the two buffers.
// Interior tiles only: M, N multiples of 256, K a multiple of 128 (the launcher checks).
// ------------------------------------------------------------------------------------------------
// Ablations run while building it (tools/gemm_w4_ablate.py, 60 back-to-back launches, 16384 x 4096 x 14336): whole kernel 1217-1236 us (8-phase
// kernel 1323, hipBLASLt 1163-1187); without the fragment reads 1050; without the DMA 1013; unswizzled (perfectly coalesced) DMA source addresses
// change nothing (1047 vs 1052) — the pieces cost what they cost because DMA and reads share the CU's LDS / texture-address ports, not because
// of their address pattern.  The ablation instantiations are gone again (they tripled the build time).
// VAR 1 (round 4, bf16 output only): the same K loop with the GENERAL epilogue (row scale, bias, activation, residual: epilogue_w4<true>) and
// an M tail: the last row tile's A rows at and past M read as zeros through the panel's buffer descriptor, its C rows are dropped by theirs.
template <bool OUT_F32, int VAR = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_256w4(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // The wave claims its SIMD's WHOLE register file (an asm clobber of v255: 256 VGPRs + 256 AGPRs), so no wave of another kernel can be placed on this
  // CU while the block runs.  Round 4 added it because with fewer registers (the general variant: 208 + 256) the output sporadically carried garbage /
  // NaNs beside the other stream's kernels; round 6 found WHY (W4_KEEP2 above: an asm v_accvgpr_read overwriting the data register of a buffer_store
  // that fetches its data late when the CU's memory pipeline is shared with foreign waves) and fixed the mechanism in the epilogues: without the claim
  // and with the fix the repro configurations are clean (0 of 1572 side-stream launches differ, IFT bench finite 4 of 4; with neither: 3 of 400, NaN
  // every run), and the step time is the same either way (PT 402.4 / 402.4 / 402.9 ms without vs 402.4 / 402.8 / 403.1 with the claim, IFT 687.8 / 686.6
  // vs 687.9 / 687.0: profiles/r06_nan_root_cause.txt).  The claim STAYS as the second line of defence: the keep-alive moves the overwrite one 16-row
  // block (hundreds of cycles) away from the store, it is not an interlock, while a CU without foreign waves fetches store data at once.
  // VP_W4_CLOBBER_MASK (bit VAR set = that instantiation claims) exists for the experiments (tools/nan_bisect_r06.sh, tools/nan_validate_r06.sh).
#ifndef VP_W4_CLOBBER_MASK
#define VP_W4_CLOBBER_MASK 7
#endif
  if constexpr ((VP_W4_CLOBBER_MASK >> VAR) & 1) asm volatile("" ::: "v255");
  bf16_t* smem = (bf16_t*)smem_raw;                    // [buf 0: A 256x64 | B 256x64][buf 1][C staging 4 x 8 KB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  const int tiles_m = VAR == 1 ? (p.M + 255) >> 8 : p.M >> 8, tiles_n = p.N >> 8;
  const int ntiles = tiles_m * tiles_n;
  const int nt = p.K >> 6;                             // even, >= 2

  // LDS-DMA geometry: 16-byte chunk q = it * 256 + tid (it 0..7) of an operand lands at LDS chunk q (lane-linear) = row q >> 3, slot q & 7;
  // it is chunk (q & 7) ^ ((row >> 1) & 7) of that row on the source side, so the 128-byte LDS rows read conflict-free with ds_read_b128
  const int drow = tid >> 3, dsw = ((tid & 7) ^ ((drow >> 1) & 7)) << 3;
  // one lane offset per operand (bytes into the 256-row panel); the piece (32 rows each) and the K-tile go into the scalar offset
  // B rows are swizzled by ((row >> 3) & 3) * 2 + ((row >> 1) & 1) instead: the fragment reads below visit them in the epilogue's column order
  const int dswb = ((tid & 7) ^ ((((drow >> 3) & 3) << 1) | ((drow >> 1) & 1))) << 3;
  const uint32_t vA = (uint32_t)(drow * p.lda + dsw) * 2u, vB = (uint32_t)(drow * p.ldb + dswb) * 2u;
  const uint32_t stepA = __builtin_amdgcn_readfirstlane((uint32_t)(32 * p.lda * 2)), stepB = __builtin_amdgcn_readfirstlane((uint32_t)(32 * p.ldb * 2));
  // fragment read addresses (bytes): + i * 2048 per 16-row block; one set per LDS buffer so the loop needs no address arithmetic
  const int fsw0 = (g ^ ((fr >> 1) & 7)) << 3, fsw1 = ((4 + g) ^ ((fr >> 1) & 7)) << 3;
  const uint32_t ldsb = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const uint32_t aad0 = ldsb + 2u * (uint32_t)((wr * 128 + fr) * 64 + fsw0), aad1 = ldsb + 2u * (uint32_t)((wr * 128 + fr) * 64 + fsw1);
  // B fragment r = 4 h + 2 s + jj, MFMA row fr  <-  tile row 64 h + 32 s + 8 (fr >> 2) + 4 jj + (fr & 3)  (see epilogue_w4): lane base row
  // 8 (fr >> 2) + (fr & 3), fragment offset W4_BOFF(r) rows; that row's swizzle key is fr >> 1 for every fragment, like the A side's
  const int brow = wc * 128 + ((fr >> 2) << 3) + (fr & 3);
  const uint32_t bad0 = ldsb + 32768u + 2u * (uint32_t)(brow * 64 + fsw0), bad1 = ldsb + 32768u + 2u * (uint32_t)(brow * 64 + fsw1);
  uint32_t aad0x = aad0 + 65536u, aad1x = aad1 + 65536u, bad0x = bad0 + 65536u, bad1x = bad1 + 65536u;
  asm volatile("" : "+v"(aad0x), "+v"(aad1x), "+v"(bad0x), "+v"(bad1x));     // keep them in registers (not re-derived inside the loop)
  // LDS destination of this wave's first piece in buffer 0 / 1 (every piece is 1 KB per wave, 4 KB per workgroup; 16 pieces = one buffer)
  const uint32_t m0b0 = __builtin_amdgcn_readfirstlane(ldsb + (uint32_t)wave * 1024u), m0b1 = m0b0 + 65536u;

  int sbm = 0, sbn = 0;
  if (gridDim.x == 256 && !(p.dbg & 0x40000)) {
    if (tiles_m % 16 == 0 && tiles_n % 16 == 0) { sbm = 16; sbn = 16; }
    else if (tiles_m % 32 == 0 && tiles_n % 8 == 0) { sbm = 32; sbn = 8; }
  }
#define TILE_OF(V) (sbm ? tile_coord_sb((V), tiles_m, tiles_n, sbm, sbn) : tile_coord_256((V), tiles_m, tiles_n))
  int v = blockIdx.x;
  TileCoord tc = TILE_OF(v);
  if ((p.dbg & 0x10000) && threadIdx.x == 0) vp_dbg_stamps[blockIdx.x * 8 + 0] = wall_clock64();
  if ((p.dbg & 0x80000) && (blockIdx.x & 3)) return;     // (measurement aid, tools/gemm_epilogue_probe.py: a quarter of the CUs work, the rest leave — results are wrong)
  // DMA stream state (wave-uniform): next K-tile to fetch = K-tile `kn` of tile `vn`; its byte offset along K is the soffset `kofs`
  typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
  auto make_rs = [&](const bf16_t* base, long rows_ld, int rows = 256) -> u32x4s {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    u32x4s r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)((uint32_t)rows * (uint32_t)rows_ld * 2u));
    r[3] = 0x00020000u;
    return r;
  };
  int vn = v, kn = 0;
  uint32_t kofs = 0, soff = 0;                         // kofs: byte offset of the stream's K-tile along K; soff: kofs + piece * step, walked by asm
#define W4_AROWS(M0) (VAR == 1 ? min(256, p.M - (M0)) : 256)
  u32x4s rsA = make_rs(p.A + (long)tc.m0 * p.lda, p.lda, W4_AROWS(tc.m0)), rsB = make_rs(p.B + (long)tc.n0 * p.ldb, p.ldb);
#define W4_DMA(VOFF, RS) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(VOFF), "s"(RS), "s"(soff) : "memory")
#define W4_SOFF0() asm volatile("s_mov_b32 %0, %1" : "=s"(soff) : "s"(kofs))                  /* first piece of an operand */
#define W4_SOFFBUMP(STEP) asm volatile("s_add_u32 %0, %0, %1" : "+s"(soff) : "s"(STEP) : "scc")
#define W4_M0SET(X) asm volatile("s_mov_b32 m0, %0" ::"s"(X) : "memory")
#define W4_M0BUMP() asm volatile("s_add_u32 m0, m0, 0x1000" ::: "memory", "scc")
#define W4_KBUMP() asm volatile("s_add_u32 %0, %0, 0x80" : "+s"(kofs)::"scc")
#define ADVANCE_STREAM()                                                                               \
  {                                                                                                    \
    if (++kn == nt) {                                                                                  \
      kn = 0;                                                                                          \
      kofs = 0;                                                                                        \
      if (vn + (int)gridDim.x < ntiles) vn += gridDim.x;      /* else: harmless in-bounds re-fetch of the last tile */ \
      const TileCoord tn_ = TILE_OF(vn);                                                               \
      rsA = make_rs(p.A + (long)tn_.m0 * p.lda, p.lda, W4_AROWS(tn_.m0));                              \
      rsB = make_rs(p.B + (long)tn_.n0 * p.ldb, p.ldb);                                                \
    }                                                                                                  \
  }
#define W4_MFMA(ACC, BF, AF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(BF), "v"(AF))
#define W4_MFMA0(ACC, BF, AF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "+a"(ACC) : "v"(BF), "v"(AF))
#define W4_LDS(DST, ADDR, OFF) do { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF)); } while (0)
#define W4_PIN(F) asm volatile("" : "+v"(F))
#define W4_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define W4_BAR() asm volatile("s_barrier" ::: "memory")
  // MFMA number m of a K-step (0..63): B fragment (m >> 3), A fragment (m & 7): consecutive MFMAs never share an accumulator
#define W4_MF1(MF, FA, FB, M) MF(acc[(M) >> 5][(M) & 7][((M) >> 3) & 3], FB[(M) >> 3], FA[(M) & 7])

  // ---- prologue: K-tiles 0 and 1 of the first output tile
  W4_M0SET(m0b0);
#pragma unroll
  for (int t = 0; t < 2; ++t) {                        // (m0 runs on from buffer 0 into buffer 1)
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      if (d == 0 || d == 8) W4_SOFF0();
      if (d < 8) { W4_DMA(vA, rsA); W4_SOFFBUMP(stepA); } else { W4_DMA(vB, rsB); W4_SOFFBUMP(stepB); }
      W4_M0BUMP();
    }
    W4_KBUMP();
    ADVANCE_STREAM();
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");    // K-tile 0 landed (this wave's part)
  W4_BAR();
  bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { W4_LDS(fa0[r], aad0, r * 2048); W4_LDS(fb0[r], bad0, vp_w4_boff(r)); }
  // asm ds_reads are invisible to the compiler's wait-count bookkeeping: if it moves or spills one of these registers before the data has
  // landed it saves garbage (seen: a scratch_store of fa0[7] right behind its ds_read, in front of the per-wave dispatch below).  Every point
  // where compiler-generated code follows in-flight fragment reads therefore waits for them first.
  W4_LGKM0();

  // One K-tile.  CUR = its LDS buffer (0 / 1: literal, the loop is unrolled by two), MF0 = the MFMA form of its K-step 0 (zero-init on the
  // first K-tile of an output tile), PH = this wave's phase.  Gap g = the slot behind MFMA g of the K-step; every memory instruction of the
  // schedule sits PH gaps later in wave PH.  Why: the four waves of the block run the same stream in lock-step behind the barriers, so without
  // the phase all four hand the SAME instruction to the CU's one texture-address unit / one LDS in the same cycle, and the last of them waits
  // three instruction times before its next MFMA can issue (rocprofv3: SQ_VMEM_TA_ADDR_FIFO_FULL 34x hipBLASLt's; an LDS-DMA piece cost ~27
  // cycles of MFMA issue, DMA and fragment reads together +24 % over either alone).  hipBLASLt's hand-written kernel does the same with two
  // copies of its loop chosen by the SIMD id; here every wave has its own slot.
#define W4_AT(M, G) ((M) - (PH_) == (G))
#define W4_KTILE(CUR, MF0, PH)                                                                         \
  {                                                                                                    \
    constexpr int PH_ = (PH);                                                                          \
    const uint32_t ra1_ = (CUR) ? aad1x : aad1, rb1_ = (CUR) ? bad1x : bad1;      /* K-step 1 of this K-tile  */ \
    const uint32_t ra0n_ = (CUR) ? aad0 : aad0x, rb0n_ = (CUR) ? bad0 : bad0x;    /* K-step 0 of the next one */ \
    W4_LGKM0();                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { W4_PIN(fa0[i]); W4_PIN(fb0[i]); }                  \
    /* ---- K-step 0: 64 MFMAs on f0 */                                                                \
    vp_static_for<64>([&](auto mc_) __attribute__((always_inline)) {                                   \
      constexpr int m = decltype(mc_)::value;                                                          \
      W4_MF1(MF0, fa0, fb0, m);                                                                        \
      constexpr int g_ = m - PH_;                                                                      \
      if constexpr (g_ >= 0 && g_ < 16 && !(g_ & 1)) W4_LDS(fa1[(g_ >> 1) & 7], ra1_, ((g_ >> 1) & 7) * 2048);   /* A fragments of K-step 1: gaps 0,2..14 */ \
      if (g_ == 18) W4_LGKM0();                                                                        \
      if (g_ == 19) W4_BAR();                                                      /* the A half of this buffer is free */ \
      if (g_ == 17) W4_SOFF0();                                                                        \
      if (g_ == 20) W4_M0SET((CUR) ? m0b1 : m0b0);                                                     \
      if (g_ >= 21 && g_ <= 56 && (g_ - 21) % 5 == 0) W4_DMA(vA, rsA);             /* A pieces of K-tile kt+2: gaps 21,26..56 */ \
      if (g_ >= 22 && g_ <= 57 && (g_ - 22) % 5 == 0) W4_M0BUMP();                                     \
      if (g_ >= 23 && g_ <= 53 && (g_ - 23) % 5 == 0) W4_SOFFBUMP(stepA);          /* 23,28..53: 7 bumps between the 8 pieces */ \
      if constexpr (g_ >= 24 && g_ <= 40 && ((g_ - 24) % 5 == 0 || (g_ - 24) % 5 == 1))    /* B fragments: gaps 24,25,29,30,..,39,40 */ \
        W4_LDS(fb1[(((g_ - 24) / 5) * 2 + (g_ - 24) % 5) & 7], rb1_, vp_w4_boff((((g_ - 24) / 5) * 2 + (g_ - 24) % 5) & 7)); \
      if (g_ == 58) W4_SOFF0();                                                    /* B operand starts at the K-tile's offset again */ \
      if (g_ == 44) W4_LGKM0();                                                                        \
      if (g_ == 45) W4_BAR();                                                      /* the B half is free */ \
    });                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { W4_PIN(fa1[i]); W4_PIN(fb1[i]); }                  \
    /* ---- K-step 1: 64 MFMAs on f1 */                                                                \
    vp_static_for<64>([&](auto mc_) __attribute__((always_inline)) {                                   \
      constexpr int m = decltype(mc_)::value;                                                          \
      W4_MF1(W4_MFMA, fa1, fb1, m);                                                                    \
      constexpr int g_ = m - PH_;                                                                      \
      if (g_ == 1 || g_ == 5) W4_DMA(vB, rsB);                                     /* B pieces 0,1 */  \
      if (g_ == 2 || g_ == 6) W4_M0BUMP();                                                             \
      if (g_ == 3 || g_ == 7) W4_SOFFBUMP(stepB);                                                      \
      if (g_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");              /* K-tile kt+1 landed (this wave's part): 10 pieces of kt+2 newer */ \
      if (g_ == 11) W4_BAR();                                                      /* ... and everybody's */ \
      if (g_ >= 16 && g_ <= 51 && (g_ - 16) % 7 == 0) W4_DMA(vB, rsB);             /* B pieces 2..7: gaps 16,23,..,51 */ \
      if (g_ >= 17 && g_ <= 45 && (g_ - 17) % 7 == 0) W4_M0BUMP();                                     \
      if (g_ >= 19 && g_ <= 47 && (g_ - 19) % 7 == 0) W4_SOFFBUMP(stepB);                              \
      if (g_ == 53) W4_KBUMP();                                                                        \
      constexpr int r_ = vp_w4_rd_slot(g_);                                        /* the 16 fragments of K-step 0 of K-tile kt+1 */ \
      if constexpr (r_ >= 0 && r_ < 8) W4_LDS(fa0[r_ & 7], ra0n_, (r_ & 7) * 2048);                    \
      if constexpr (r_ >= 8) W4_LDS(fb0[r_ & 7], rb0n_, vp_w4_boff(r_ & 7));                              \
    });                                                                                                \
    W4_LGKM0();          /* the fa0 / fb0 reads issued above are in flight and ADVANCE_STREAM is compiler-generated code: wait first (the invariant \
                            stated at the prologue; the reads are >= 15 MFMAs old here, so the wait is already satisfied in practice) */ \
    ADVANCE_STREAM();                                                                                  \
  }
#define W4_MAINLOOP(PH)                                                                                \
  {                                                                                                    \
    W4_KTILE(0, W4_MFMA0, PH);                                                                         \
    W4_KTILE(1, W4_MFMA, PH);                                                                          \
    for (int kt = 2; kt < nt; kt += 2) {                                                               \
      W4_KTILE(0, W4_MFMA, PH);                                                                        \
      W4_KTILE(1, W4_MFMA, PH);                                                                        \
    }                                                                                                  \
  }

  // The 256 accumulators are ONE value chain for the whole kernel: defined once here, then only ever modified in place ("+a") — also by the
  // zero-initialising MFMA form of each output tile's first K-step.  With a fresh definition ("=a") per output tile the register allocator
  // bridged the two definitions with ~500 v_accvgpr_mov per tile at the K loop's entry and exit (epilogue 10 us instead of 4).
  f32x4 acc[2][8][4];                                  // [B fragment >> 2][A fragment][B fragment & 3]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "=a"(acc[h][i][j]));
  while (true) {
    const TileCoord tcur = tc;
    if ((p.dbg & 0x10000) && threadIdx.x == 0 && v == (int)blockIdx.x) {
      vp_dbg_stamps[blockIdx.x * 8 + 1] = wall_clock64();
      vp_dbg_stamps[blockIdx.x * 8 + 6] = clock64();
    }
    W4_MAINLOOP(0)          // (per-wave phases 0..3 — hipBLASLt runs two copies of its loop one MFMA apart, chosen by the SIMD id — measured: no gain
                            // here once the pieces are spread out, 1219-1241 us at 16384 x 4096 x 14336 either way; one copy is a quarter of the code)
    if ((p.dbg & 0x10000) && threadIdx.x == 0 && v == (int)blockIdx.x) {
      vp_dbg_stamps[blockIdx.x * 8 + 2] = wall_clock64();
      vp_dbg_stamps[blockIdx.x * 8 + 7] = clock64();
    }
    W4_LGKM0();                                        // (the reads the last K-tile issued for the next tile's first fragments have landed)
    // The fragment registers are NOT kept across the epilogue: with all 256 AGPRs holding accumulators the epilogue's own registers (staging
    // addresses, two slots of prefetched residual / gate|up rows) did not fit beside 64 live fragment VGPRs and the allocator spilled
    // accumulator tuples to scratch.  They are re-read (16 ds_read_b128, ~300 cycles) once the tile is stored.
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" : "=v"(fa0[i])); asm volatile("" : "=v"(fb0[i])); }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA results visible to the compiler's v_accvgpr_reads
    if (OUT_F32) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[h][i][j]));
    }
    const int vnext = v + gridDim.x;
    const int mr = tcur.m0 + wr * 128, nc = tcur.n0 + wc * 128;
    if (!OUT_F32) {
      epilogue_w4<VAR>(p, acc, mr, nc, lane);        // (the launcher sends bias / activation epilogues and unaligned C to the 8-phase kernel:
                                                       // the general epilogue's registers beside 256 accumulators made the allocator spill AGPRs)
    } else {
      float* c = (float*)p.C + (long)(mr + fr) * p.ldc + nc + g * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) *(f32x4*)(c + (long)(i * 16) * p.ldc + h * 64 + (j >> 1) * 32 + (j & 1) * 4) = acc[h][i][j];
    }
    if ((p.dbg & 0x10000) && threadIdx.x == 0 && v == (int)blockIdx.x) {
      vp_dbg_stamps[blockIdx.x * 8 + 3] = wall_clock64();          // epilogue issued
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      vp_dbg_stamps[blockIdx.x * 8 + 4] = wall_clock64();          // ... and its stores drained
    }
    if (vnext >= ntiles) break;
    v = vnext;
    tc = TILE_OF(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) { W4_LDS(fa0[r], aad0, r * 2048); W4_LDS(fb0[r], bad0, vp_w4_boff(r)); }      // K-tile 0 of the next tile: buffer 0 (nt is even)
    W4_LGKM0();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing dummy DMAs must not outlive the workgroup's LDS
  if ((p.dbg & 0x10000) && threadIdx.x == 0) vp_dbg_stamps[blockIdx.x * 8 + 5] = wall_clock64();
  if ((p.dbg & 0x10000) && threadIdx.x == 0 && false) vp_dbg_stamps[blockIdx.x * 8 + 0] = 0;
#undef TILE_OF
#undef W4_AROWS
#undef W4_DMA
#undef W4_M0SET
#undef W4_M0BUMP
#undef W4_SOFF0
#undef W4_SOFFBUMP
#undef W4_KBUMP
#undef ADVANCE_STREAM
#undef W4_MFMA
#undef W4_MFMA0
#undef W4_LDS
#undef W4_PIN
#undef W4_LGKM0
#undef W4_BAR
#undef W4_MF1
#undef W4_KTILE
#undef W4_MAINLOOP
#undef W4_AT
}

// ------------------------------------------------------------------------------------------------
// generic path: any M, N, K, any alignment (register-staged, zero-filled K tail). 64x64x32 tile.
// ------------------------------------------------------------------------------------------------
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_generic(GemmArgs p) {
  constexpr int LD = 40;   // 32 + 8 pad (80-B rows: 16-B aligned, conflict-free b128 reads)
  __shared__ __attribute__((aligned(16))) bf16_t As[64 * LD];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + 63) >> 6;
  const int m0 = (blockIdx.x / tiles_n) << 6, n0 = (blockIdx.x % tiles_n) << 6;
  const int srow = tid >> 2, skc = (tid & 3) * 8;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, g = lane >> 4;
  f32x4 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    bf16_t a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + skc + e;
      a[e] = (m0 + srow < p.M && k < p.K) ? p.A[(long)(m0 + srow) * p.lda + k] : (bf16_t)0;
      b[e] = (n0 + srow < p.N && k < p.K) ? p.B[(long)(n0 + srow) * p.ldb + k] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      As[srow * LD + skc + e] = a[e];
      Bs[srow * LD + skc + e] = b[e];
    }
    __syncthreads();
    bf16x8 xf[2], wf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xf[i] = *(const bf16x8*)(As + (wm * 32 + i * 16 + fr) * LD + g * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[j] = *(const bf16x8*)(Bs + (wn * 32 + j * 16 + fr) * LD + g * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store4<OUT_F32>(p, m0 + wm * 32 + i * 16 + fr, n0 + wn * 32 + j * 16 + g * 4, acc[i][j]);
}

// ------------------------------------------------------------------------------------------------
// 2-D transpose (bf16): out[c][r] = in[r][c]; 64x64 tiles through LDS.  Interior tiles of 16-byte-aligned matrices move as
// 16-byte vectors on both sides (row pitch 66 elements keeps the column gathers conflict-free); edge / unaligned tiles go element-wise.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* in, bf16_t* out, int rows, int cols, long ldi,
                                                             long ldo, long bsi = 0, long bso = 0) {
  __shared__ bf16_t t[64][66];
  in += (long)blockIdx.z * bsi;                        // batched form (vp_transpose_batched_bf16): matrix blockIdx.z of the batch
  out += (long)blockIdx.z * bso;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bool vec = r0 + 64 <= rows && c0 + 64 <= cols && (ldi & 7) == 0 && (ldo & 7) == 0 &&
                   ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
  if (vec) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = it * 256 + threadIdx.x, r = q >> 3, ch = q & 7;
      const bf16x8 v = *(const bf16x8*)(in + (long)(r0 + r) * ldi + c0 + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; e += 2) *(uint32_t*)&t[r][ch * 8 + e] = (uint32_t)(uint16_t)v[e] | ((uint32_t)(uint16_t)v[e + 1] << 16);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = it * 256 + threadIdx.x, c = q >> 3, rc = q & 7;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (short)t[rc * 8 + e][c];
      *(bf16x8*)(out + (long)(c0 + c) * ldo + r0 + rc * 8) = o;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < rows && c < cols) ? in[(long)r * ldi + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(long)c * ldo + r] = t[tx][i];
  }
}

// Dynamic tile scheduling of the persistent 8-phase kernels: a launch claims its tiles from per-XCD counters when the caller hands it a
// counter block (`sched_ws`: 16 ints of caller-owned device memory, zeroed once; the kernel leaves it zeroed, so back-to-back launches on one
// stream share one block), else the static round-robin walk.  The library owns no device memory and keeps no pointer (SURVEY 8b "Ownership").
__global__ void occupy_kernel(long cycles) {
  extern __shared__ unsigned char occ_lds[];
  occ_lds[threadIdx.x] = 1;
  const long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (occ_lds[threadIdx.x] == 77) occ_lds[0] = 2;
}

static bool vp_c_nt_enabled() {                        // VP_GEMM_C_NT=0 switches the non-temporal C stores off (A/B)
  static int e = -1;
  if (e < 0) { const char* v = getenv("VP_GEMM_C_NT"); e = v ? atoi(v) : 1; }
  return e != 0;
}

static bool vp_ph4_enabled() {
  static int e = -1;
  // default ON since round 2: in-step A/B on one box, alternating runs: 458.1 / 459.0 ms per step against 461.7 / 462.5 with the 8-phase loop
  // (GEMM 1337 vs 1324 TFLOP/s), bit-identical results; VP_GEMM_PH4=0 selects the 8-phase loop
  if (e < 0) { const char* v = getenv("VP_GEMM_PH4"); e = v ? atoi(v) : 1; }
  return e != 0;
}

extern "C" {

long vp_gemm_sched_workspace_bytes(void) { return 64; }

#ifdef VP_DEBUG
// dev aid (tools/gemm_interference.py): `blocks` workgroups that each pin 64 KB of LDS (so no 8-phase GEMM block fits beside them) and spin
// for `cycles` shader cycles -- a stand-in for a collective kernel holding CUs
int vp_debug_occupy(int blocks, long cycles, hipStream_t stream) {
  VP_REQUIRE(blocks > 0 && cycles > 0, VP_ERR_BAD_ARG, "vp_debug_occupy: bad args");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; }
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), 65536, stream, cycles);
  return vp_check_launch("vp_debug_occupy");
}
#endif

#ifdef VP_DEBUG
static int g_gemm_dbg = -1;                           // VP_GEMM_DBG, or vp_debug_gemm_flags(): measurement flags exist in -DVP_DEBUG builds only
static int vp_gemm_dbg() {
  if (g_gemm_dbg < 0) { const char* e = getenv("VP_GEMM_DBG"); g_gemm_dbg = e ? atoi(e) : 0; }
  return g_gemm_dbg;
}
#else
static constexpr int vp_gemm_dbg() { return 0; }      // the sealed library reads no debug switch (a stray VP_GEMM_DBG cannot reach a kernel)
#endif
#ifdef VP_DEBUG
// measurement aid (bench.py: shader clock under load): 0x10000 = the persistent kernel's first tile writes wall-clock / shader-cycle stamps
int vp_debug_gemm_flags(int flags) { g_gemm_dbg = flags; return 0; }

int vp_debug_stamps(long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(vp_dbg_stamps), sizeof(long) * 256 * 8) == hipSuccess ? 0 : 1;
}
#endif

int vp_gemm_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                 const void* bias, const void* residual, long ldr, int epilogue, int out_f32, int force_generic, int* sched_ws,
                 hipStream_t stream) {
  VP_REQUIRE(M > 0 && N > 0 && K > 0, VP_ERR_BAD_ARG, "vp_gemm_bf16: non-positive dims %d %d %d", M, N, K);
  VP_REQUIRE(A && B && C, VP_ERR_BAD_ARG, "vp_gemm_bf16: null operand");
  VP_REQUIRE(lda >= K && ldb >= K && ldc >= N, VP_ERR_BAD_ARG, "vp_gemm_bf16: leading dims too small");
  VP_REQUIRE((epilogue & 0xff) >= 0 && (epilogue & 0xff) <= 3, VP_ERR_BAD_ARG, "vp_gemm_bf16: bad epilogue %d", epilogue);
  VP_REQUIRE(force_generic == 0 || force_generic == 1 || force_generic == 2 || force_generic == 3 || force_generic == 7 || force_generic == 8 ||
                 force_generic == 13 || force_generic == 14, VP_ERR_BAD_ARG, "vp_gemm_bf16: unknown kernel selector %d", force_generic);
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)residual, M, N, K,
             lda, ldb, ldc, ldr, epilogue, 0, 0, nullptr, 0, nullptr, 0};
  p.dbg = vp_gemm_dbg();
  const bool fast = (force_generic != 1) && (K % 64 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                    ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
  const long big_tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  {
    static int w4_env = -1;
    if (w4_env < 0) { const char* e = getenv("VP_GEMM_W4"); w4_env = e ? atoi(e) : 1; }      // default since round 3 (VP_GEMM_W4=0: the 8-phase kernel)
    const bool w4_ok = fast && M % 256 == 0 && N % 256 == 0 && K % 128 == 0 && (big_tiles >= 192 || force_generic == 8) &&
                       !bias && (epilogue & 0xff) == EPI_NONE && (((uintptr_t)C) & 15) == 0 &&
                       (out_f32 ? (!residual && ldc % 4 == 0)
                                : (ldc % 8 == 0 && ldc < (1L << 22) &&                       // (32-bit byte offsets inside a 128-row sub-tile)
                                   (!residual || (ldr % 8 == 0 && ldr < (1L << 22) && (((uintptr_t)residual) & 15) == 0))));
    // The 4-wave kernel walks its tiles statically.  Next to RCCL kernels (world > 1) a CU that a collective holds delays that block's share
    // (tools/gemm_interference.py: up to 1.45x for the launches that overlap a collective); in the PT step that is the handful of GEMMs under the
    // 0.2 GB gradient all-reduce, against 6-10 % on every launch with the 8-phase kernel and its per-XCD tile claims — so the multi-GPU step uses
    // it too (VP_GEMM_W4=0 / VP_GEMM_W4=2 "only when no collective can run beside it" select the 8-phase kernel).
    if (w4_ok && (force_generic == 8 || (force_generic == 0 && (w4_env == 1 || (w4_env == 2 && !sched_ws))))) {
      p.c_nt = (!out_f32 && N <= 8192 && vp_c_nt_enabled()) ? 1 : 0;
      static bool attr_w4 = false;
      if (!attr_w4) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr_w4 = true;
      }
      const unsigned g4 = (unsigned)(big_tiles > 256 ? 256 : big_tiles);
      if (out_f32) hipLaunchKernelGGL(gemm_nt_256w4<true>, dim3(g4), dim3(256), 163840, stream, p);
      else hipLaunchKernelGGL(gemm_nt_256w4<false>, dim3(g4), dim3(256), 163840, stream, p);
      return vp_check_launch("vp_gemm_bf16");
    }
  }
  {
    // General variant of the 4-wave kernel (round 4; force code 14; VP_GEMM_W4G=0 turns the automatic choice off): bias / activation / residual /
    // row-scale epilogues and an M tail, for the tower, projector, head and DPT launches that used to fall to the 8-phase kernel.
    static int w4g_env = -1;
    if (w4g_env < 0) { const char* e = getenv("VP_GEMM_W4G"); w4g_env = e ? atoi(e) : 1; }
    const int epi8 = epilogue & 0xff;
    const bool w4g_ok = fast && !out_f32 && N % 256 == 0 && K % 128 == 0 && M >= 256 && (((uintptr_t)C) & 15) == 0 && ldc % 8 == 0 && ldc < (1L << 22) &&
                        (!bias || (((uintptr_t)bias) & 15) == 0) && !(residual && epi8 != EPI_NONE) &&
                        (!residual || (ldr % 8 == 0 && ldr < (1L << 22) && (((uintptr_t)residual) & 15) == 0));
    // erf-GELU launches stay on the 8-phase kernel: ~50 VALU instructions per element at one wave per SIMD cost more than the K loop gains
    // (ConvNeXt fc1 18432 x 6144 x 1536: 861 -> 800 TF/s; bias / residual launches: 1092 -> 1363, 1062 -> 1440, 892 -> 1101 TF/s)
    if (w4g_ok && (force_generic == 14 || (force_generic == 0 && w4g_env == 1 && big_tiles >= 64 && epi8 != EPI_GELU))) {     // (tools/vit_w4_probe.py: ahead of the 128-tile kernel from 76 tiles on)
      static bool attr_w4g = false;
      if (!attr_w4g) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr_w4g = true;
      }
      const unsigned g4 = (unsigned)(big_tiles > 256 ? 256 : big_tiles);
      hipLaunchKernelGGL((gemm_nt_256w4<false, 1>), dim3(g4), dim3(256), 163840, stream, p);
      return vp_check_launch("vp_gemm_bf16");
    }
  }
  if (fast && (force_generic == 7 || (force_generic == 0 && big_tiles >= 192 && M >= 256 && N >= 256))) {   // default large-problem kernel
    static bool attr_p8 = false;
    if (!attr_p8) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_256p8<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute((const void*)gemm_nt_256p8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      attr_p8 = true;
    }
    // persistent (one block per CU streaming its output tiles) when the K-tile count is even (buffer parity is then the same
    // for every output tile); otherwise one block per output tile
    const unsigned g8 = (unsigned)(((K / 64) % 2 == 0 && big_tiles > 256 && !(p.dbg & 0x20000)) ? 256 : big_tiles);
    if (g8 == 256) p.sched = sched_ws;
    p.c_nt = (!out_f32 && N <= 8192 && (long)M * ldc * 2 < 0x7fffffffL && vp_c_nt_enabled()) ? 1 : 0;
    if (out_f32) hipLaunchKernelGGL(gemm_nt_256p8<true>, dim3(g8), dim3(512), 131072, stream, p);
    else if (vp_ph4_enabled() && force_generic == 0) {
      static bool attr_p4 = false;
      if (!attr_p4) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256p8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr_p4 = true;
      }
      hipLaunchKernelGGL((gemm_nt_256p8<false, true>), dim3(g8), dim3(512), 131072, stream, p);
    } else hipLaunchKernelGGL(gemm_nt_256p8<false>, dim3(g8), dim3(512), 131072, stream, p);
  } else if (fast && force_generic == 13 && !out_f32) {       // 4-phase variant of the 8-phase kernel (A/B testing)
    (void)hipFuncSetAttribute((const void*)gemm_nt_256p8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const unsigned g8 = (unsigned)(((K / 64) % 2 == 0 && big_tiles > 256 && !(p.dbg & 0x20000)) ? 256 : big_tiles);
    hipLaunchKernelGGL((gemm_nt_256p8<false, true>), dim3(g8), dim3(512), 131072, stream, p);
  } else if (fast && force_generic != 2 && (force_generic == 3 || (big_tiles >= 192 && M >= 256 && N >= 256))) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_256<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute((const void*)gemm_nt_256<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      attr_done = true;
    }
    const unsigned pgrid = (unsigned)(big_tiles < 256 ? big_tiles : 256);      // persistent: one block per CU
    if (out_f32) hipLaunchKernelGGL(gemm_nt_256<true>, dim3(pgrid), dim3(512), 131072, stream, p);
    else hipLaunchKernelGGL(gemm_nt_256<false>, dim3(pgrid), dim3(512), 131072, stream, p);
  } else if (fast) {
    const int grid = ((M + 127) / 128) * ((N + 127) / 128);
    if (out_f32) hipLaunchKernelGGL(gemm_nt_128<true>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_nt_128<false>, dim3(grid), dim3(256), 0, stream, p);
  } else {
    const int grid = ((M + 63) / 64) * ((N + 63) / 64);
    if (out_f32) hipLaunchKernelGGL(gemm_nt_generic<true>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_nt_generic<false>, dim3(grid), dim3(256), 0, stream, p);
  }
  return vp_check_launch("vp_gemm_bf16");
}

// Residual GEMM that also emits the statistics of the next RMSNorm (reference: HF LlamaDecoderLayer.forward, modeling_llama.py: residual +
// o_proj(...) / residual + mlp(...), each followed by an RMSNorm): C[M,N] = A B^T + residual (bf16, as vp_gemm_bf16) and
// sumsq_part[M, N / 16] (fp32) = per-row sums of squares of the result over 16-column groups (vp_rstd_from_sumsq adds them in a fixed order).
// Lean one-wave-per-SIMD kernel only: M, N multiples of 256, K of 128, 16-byte aligned rows; else VP_ERR_UNSUPPORTED_SHAPE.
int vp_gemm_bf16_sumsq(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, const void* residual, long ldr,
                       float* sumsq_part, hipStream_t stream) {
  VP_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && residual && sumsq_part, VP_ERR_BAD_ARG, "vp_gemm_bf16_sumsq: bad operands");
  VP_REQUIRE(lda >= K && ldb >= K && ldc >= N && ldr >= N, VP_ERR_BAD_ARG, "vp_gemm_bf16_sumsq: leading dims too small");
  VP_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldr % 8 == 0 && ldc < (1L << 22) &&
                 ldr < (1L << 22) && ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)residual)) & 15) == 0,
             VP_ERR_UNSUPPORTED_SHAPE, "vp_gemm_bf16_sumsq: needs M, N %% 256 == 0, K %% 128 == 0, 16-byte aligned rows (got %d %d %d)", M, N, K);
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, (const bf16_t*)residual, M, N, K, lda, ldb, ldc, ldr, EPI_NONE, 0, 0, nullptr, 0, nullptr, 0};
  p.dbg = vp_gemm_dbg();
  p.sumsq_part = sumsq_part;
  p.c_nt = (N <= 8192 && vp_c_nt_enabled()) ? 1 : 0;
  static bool attr_w4 = false;
  if (!attr_w4) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr_w4 = true;
  }
  const long big_tiles = (long)(M / 256) * (N / 256);
  hipLaunchKernelGGL((gemm_nt_256w4<false, 2>), dim3((unsigned)(big_tiles > 256 ? 256 : big_tiles)), dim3(256), 163840, stream, p);
  return vp_check_launch("vp_gemm_bf16_sumsq");
}

// QKV projection of a decoder layer with rotate-half RoPE in the epilogue (reference: HF LlamaAttention.forward, modeling_llama.py: q/k/v_proj, then
// apply_rotary_pos_emb on q and k): C[M,N] = (row_scale (.) A) B^T, columns < rope_cols (whole 128-wide heads) rotated like vp_rope (same rounding
// points: bit-identical to vp_gemm_bf16 followed by vp_rope).  row_scale (fp32 [M], may be NULL) multiplies the fp32 accumulator before the bf16
// rounding: RMSNorm's 1/rms when gamma is folded into the frozen weight.  General variant of the one-wave-per-SIMD kernel only: head_dim 128,
// N a multiple of 256, K of 128, M >= 256, 16-byte aligned rows; anything else is VP_ERR_UNSUPPORTED_SHAPE (the caller then runs the two calls).
int vp_gemm_bf16_rope(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, const float* row_scale,
                      int rope_cols, const float* cos_t, const float* sin_t, const int* pos, int S, hipStream_t stream) {
  VP_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && cos_t && sin_t && S > 0, VP_ERR_BAD_ARG, "vp_gemm_bf16_rope: bad operands");
  VP_REQUIRE(lda >= K && ldb >= K && ldc >= N && rope_cols >= 0 && rope_cols <= N, VP_ERR_BAD_ARG, "vp_gemm_bf16_rope: leading dims / rope_cols");
  const long big_tiles = (long)((M + 255) / 256) * (N / 256);
  VP_REQUIRE(N % 256 == 0 && K % 128 == 0 && M >= 256 && rope_cols % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldc < (1L << 22) &&
                 ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)cos_t) | ((uintptr_t)sin_t)) & 15) == 0,
             VP_ERR_UNSUPPORTED_SHAPE, "vp_gemm_bf16_rope: needs N %% 256 == 0, K %% 128 == 0, M >= 256, rope_cols %% 128 == 0, 16-byte aligned rows (got %d %d %d)", M, N, K);
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, EPI_NONE, 0, 0, nullptr, 0, nullptr, 0};
  p.dbg = vp_gemm_dbg();
  p.rowscale = row_scale;
  p.rope_cos = cos_t; p.rope_sin = sin_t; p.rope_pos = pos; p.rope_S = S; p.rope_cols = rope_cols;
  static bool attr_w4g = false;
  if (!attr_w4g) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr_w4g = true;
  }
  hipLaunchKernelGGL((gemm_nt_256w4<false, 1>), dim3((unsigned)(big_tiles > 256 ? 256 : big_tiles)), dim3(256), 163840, stream, p);
  return vp_check_launch("vp_gemm_bf16_rope");
}

// Fused SwiGLU GEMMs for the decoder MLP (reference: HF LlamaMLP.forward, modeling_llama.py — down(act(gate(x)) * up(x)); the
// fused gate/up weight keeps its columns interleaved in 8-wide chunks: g0..7 | u0..7 | g8..15 | ...).
//   mode 1 (forward):  C[M,N] = A[M,K] B[N,K]^T (gate_up), C2[M,N/2] = silu(gate) * up         (aux unused)
//                      aux (optional) = fp32 [M] row scale on the accumulators: gate_up = bf16(acc * scale) (RMSNorm folded into the weight)
//   mode 2 (backward): d_act[M,N] = A B^T stays on chip; aux = gate_up[M,2N]; C[M,2N] = d_gate_up  (C2 unused)
// Only the 8-phase kernel implements these epilogues: M, N multiples of 256, K a multiple of 64, 16-byte aligned rows.
int vp_gemm_bf16_swiglu(int mode, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                        void* C2, long ldc2, const void* aux, long ldaux, int* sched_ws, hipStream_t stream) {
  VP_REQUIRE(mode == 1 || mode == 2, VP_ERR_BAD_ARG, "vp_gemm_bf16_swiglu: mode %d", mode);
  VP_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, VP_ERR_BAD_ARG, "vp_gemm_bf16_swiglu: bad operands");
  VP_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 64 == 0, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_gemm_bf16_swiglu: needs M, N multiples of 256 and K a multiple of 64 (got %d %d %d)", M, N, K);
  VP_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
                 ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C)) & 15) == 0,
             VP_ERR_BAD_ARG, "vp_gemm_bf16_swiglu: leading dims / alignment");
  if (mode == 1)
    VP_REQUIRE(C2 && ldc >= N && ldc2 >= N / 2 && ldc2 % 8 == 0 && (((uintptr_t)C2) & 15) == 0, VP_ERR_BAD_ARG,
               "vp_gemm_bf16_swiglu: forward outputs");
  if (mode == 2)
    VP_REQUIRE(aux && ldc >= 2 * N && ldaux >= 2 * N && ldaux % 8 == 0 && (((uintptr_t)aux) & 15) == 0, VP_ERR_BAD_ARG,
               "vp_gemm_bf16_swiglu: backward operands");
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, EPI_NONE, 0,
             mode, C2, ldc2, (const bf16_t*)aux, ldaux};
  p.dbg = vp_gemm_dbg();
  if (mode == 1 && aux) {                              // forward: aux = fp32 [M] row scale (one-wave-per-SIMD kernel only, checked below)
    p.rowscale = (const float*)aux;
    p.aux = nullptr;
  }
  const long big_tiles = (long)(M / 256) * (N / 256);
  {
    static int w4_env = -1;
    if (w4_env < 0) { const char* e = getenv("VP_GEMM_W4"); w4_env = e ? atoi(e) : 1; }
    if ((w4_env == 1 || (w4_env == 2 && !sched_ws)) && K % 128 == 0 && big_tiles >= 192 && ldc < (1L << 21) && ldc2 < (1L << 21) && ldaux < (1L << 21)) {   // same routing as vp_gemm_bf16
      static bool attr_w4 = false;
      if (!attr_w4) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr_w4 = true;
      }
      if (p.rowscale) {
        static bool attr_w4f = false;
        if (!attr_w4f) {
          (void)hipFuncSetAttribute((const void*)gemm_nt_256w4<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
          attr_w4f = true;
        }
        hipLaunchKernelGGL((gemm_nt_256w4<false, 2>), dim3((unsigned)(big_tiles > 256 ? 256 : big_tiles)), dim3(256), 163840, stream, p);
      } else hipLaunchKernelGGL(gemm_nt_256w4<false>, dim3((unsigned)(big_tiles > 256 ? 256 : big_tiles)), dim3(256), 163840, stream, p);
      return vp_check_launch("vp_gemm_bf16_swiglu");
    }
  }
  VP_REQUIRE(!p.rowscale, VP_ERR_UNSUPPORTED_SHAPE, "vp_gemm_bf16_swiglu: the row scale exists on the one-wave-per-SIMD kernel only (K %% 128 == 0, >= 192 tiles; got %d %d %d)", M, N, K);
  static bool attr_p8 = false;
  if (!attr_p8) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_256p8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_p8 = true;
  }
  const unsigned g8 = (unsigned)(((K / 64) % 2 == 0 && big_tiles > 256) ? 256 : big_tiles);
  if (g8 == 256) p.sched = sched_ws;
  if (vp_ph4_enabled()) {
    static bool attr_p4 = false;
    if (!attr_p4) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_256p8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      attr_p4 = true;
    }
    hipLaunchKernelGGL((gemm_nt_256p8<false, true>), dim3(g8), dim3(512), 131072, stream, p);
  } else hipLaunchKernelGGL(gemm_nt_256p8<false>, dim3(g8), dim3(512), 131072, stream, p);
  return vp_check_launch("vp_gemm_bf16_swiglu");
}

// C[M,N] (+)= A[K,M]^T B[K,N]: both operands contraction-major, i.e. the weight gradient dW[out,in] = dY[tokens,out]^T X[tokens,in]
// straight from the activation buffers (reference: autograd of nn.Linear, grad_weight = grad_output.t() @ input).  8-phase TN kernel
// only: M, N multiples of 256, K a multiple of 64, 16-byte aligned rows; anything else is VP_ERR_UNSUPPORTED_SHAPE (the caller then
// transposes and uses vp_gemm_bf16).  accumulate != 0 (fp32 output only) adds into C.
int vp_gemm_tn_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int out_f32,
                    int accumulate, int* sched_ws, hipStream_t stream) {
  VP_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, VP_ERR_BAD_ARG, "vp_gemm_tn_bf16: bad operands");
  VP_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 64 == 0, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_gemm_tn_bf16: needs M, N multiples of 256 and K a multiple of 64 (got %d %d %d)", M, N, K);
  VP_REQUIRE(lda >= M && ldb >= N && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
                 ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C)) & 15) == 0,
             VP_ERR_BAD_ARG, "vp_gemm_tn_bf16: leading dims / alignment");
  VP_REQUIRE(!accumulate || out_f32, VP_ERR_BAD_ARG, "vp_gemm_tn_bf16: accumulate needs fp32 output");
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, EPI_NONE, 0,
             accumulate ? 3 : 0, nullptr, 0, nullptr, 0};
  static bool attr_tn = false;
  if (!attr_tn) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_256p8<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)gemm_tn_256p8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_tn = true;
  }
  const long big_tiles = (long)(M / 256) * (N / 256);
  const unsigned g8 = (unsigned)(((K / 64) % 2 == 0 && big_tiles > 256) ? 256 : big_tiles);
  if (g8 == 256) p.sched = sched_ws;
  if (out_f32) hipLaunchKernelGGL(gemm_tn_256p8<true>, dim3(g8), dim3(512), 131072, stream, p);
  else hipLaunchKernelGGL(gemm_tn_256p8<false>, dim3(g8), dim3(512), 131072, stream, p);
  return vp_check_launch("vp_gemm_tn_bf16");
}


int vp_transpose_bf16(int rows, int cols, const void* in, long ld_in, void* out, long ld_out, hipStream_t stream) {
  VP_REQUIRE(rows > 0 && cols > 0 && in && out, VP_ERR_BAD_ARG, "vp_transpose_bf16: bad args");
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, rows, cols,
                     ld_in, ld_out, 0L, 0L);
  return vp_check_launch("vp_transpose_bf16");
}

int vp_transpose_batched_bf16(int batch, int rows, int cols, const void* in, long batch_stride_in, long ld_in, void* out, long batch_stride_out,
                              long ld_out, hipStream_t stream) {
  VP_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && cols > 0 && in && out, VP_ERR_BAD_ARG, "vp_transpose_batched_bf16: bad args");
  VP_REQUIRE(batch_stride_in % 8 == 0 && batch_stride_out % 8 == 0, VP_ERR_BAD_ARG, "vp_transpose_batched_bf16: batch strides must be multiples of 8 elements");
  dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out,
                     batch_stride_in, batch_stride_out);
  return vp_check_launch("vp_transpose_batched_bf16");
}

}  // extern "C"
