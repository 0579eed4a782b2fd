// bf16 MFMA GEMM for the VisPer-LM hot path:  C[M,N] = epi(A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]
// Both operands are K-contiguous ("NT"): activations [tokens, features] x nn.Linear weights [out, in].
// (dgrad uses the pre-transposed frozen weight, wgrad uses explicitly transposed operands — DESIGN.md.)
//
// Fast kernel: 128x128x64 block tile, 4 waves (2x2, each 64x64 = 4x4 MFMA 16x16x32 tiles), operands staged
// HBM -> LDS with 16-byte global_load_lds (no VGPR round trip).  LDS image is lane-linear (a glds
// constraint), so bank conflicts are removed by XOR-swizzling the 16-B chunk index on the *source*
// address and again on the ds_read_b128 (chunk ^= (row>>1)&7).  MFMA roles are swapped (A-operand =
// weight rows, B-operand = token rows) so each lane ends up with 4 consecutive output columns of one
// token row -> 8-byte bf16 stores.  Block ids are remapped XCD-aware + grouped so tiles sharing an
// operand panel sit in one XCD's L2.
#include "common.h"

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
};

__device__ __forceinline__ float apply_epi(float v, int epi) {
  switch (epi) {
    case EPI_GELU: return gelu_erf(v);
    case EPI_QUICK_GELU: return quick_gelu(v);
    case EPI_RELU: return fmaxf(v, 0.f);
    default: return v;
  }
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
      if (p.epi != EPI_NONE) {
        x = apply_epi(x, p.epi);
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
// large-problem path: 256x256x64 block tile, 8 waves (2 x 4, each 128x64 = 8x4 MFMA tiles), 2 LDS buffers
// (2 x 64 KB): tile t+1 is DMA'd into the other buffer at the START of computing tile t, so its
// global_load_lds has a whole tile of MFMA work to land under and the single barrier per K-tile (which
// carries vmcnt(0)) never stalls on it.  1 block / CU (128 KB LDS), 2 waves / SIMD.
// ------------------------------------------------------------------------------------------------
template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_nt_256(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;                    // [buf 0: A 256x64 | B 256x64][buf 1: ...]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP_M = 8;
  const int width = GROUP_M * tiles_n;
  const int group = id / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (id % width) % gsz;
  const int tn = (id % width) / gsz;
  const int m0 = tm << 8, n0 = tn << 8;

  const bf16_t* srcA[4];
  const bf16_t* srcB[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = it * 512 + tid;
    const int row = q >> 3;
    const int gc = (q & 7) ^ ((row >> 1) & 7);
    srcA[it] = p.A + (long)min(m0 + row, p.M - 1) * p.lda + gc * 8;
    srcB[it] = p.B + (long)min(n0 + row, p.N - 1) * p.ldb + gc * 8;
  }
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, g = lane >> 4;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = p.K >> 6;
#define ISSUE_TILE(T, BUF)                                                         \
  {                                                                                \
    bf16_t* As_ = smem + (BUF) * 32768;                                            \
    bf16_t* Bs_ = As_ + 16384;                                                     \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                             \
      GLDS16(srcA[it] + (long)(T) * 64, As_ + (it * 512 + wave * 64) * 8);         \
      GLDS16(srcB[it] + (long)(T) * 64, Bs_ + (it * 512 + wave * 64) * 8);         \
    }                                                                              \
  }
  ISSUE_TILE(0, 0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) ISSUE_TILE(t + 1, cur ^ 1);
    const bf16_t* As = smem + cur * 32768;
    const bf16_t* Bs = As + 16384;
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
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#undef ISSUE_TILE
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      store4<OUT_F32>(p, m0 + wr * 128 + i * 16 + fr, n0 + wc * 64 + j * 16 + g * 4, acc[i][j]);
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
// 2-D transpose (bf16): out[c][r] = in[r][c]; 64x64 tiles through LDS
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* in, bf16_t* out, int rows, int cols, long ldi,
                                                             long ldo) {
  __shared__ bf16_t t[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
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

extern "C" {

int vp_gemm_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                 const void* bias, const void* residual, long ldr, int epilogue, int out_f32, int force_generic,
                 hipStream_t stream) {
  VP_REQUIRE(M > 0 && N > 0 && K > 0, VP_ERR_BAD_ARG, "vp_gemm_bf16: non-positive dims %d %d %d", M, N, K);
  VP_REQUIRE(A && B && C, VP_ERR_BAD_ARG, "vp_gemm_bf16: null operand");
  VP_REQUIRE(lda >= K && ldb >= K && ldc >= N, VP_ERR_BAD_ARG, "vp_gemm_bf16: leading dims too small");
  VP_REQUIRE(epilogue >= 0 && epilogue <= 3, VP_ERR_BAD_ARG, "vp_gemm_bf16: bad epilogue %d", epilogue);
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)residual, M, N, K,
             lda, ldb, ldc, ldr, epilogue};
  const bool fast = (force_generic != 1) && (K % 64 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                    ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
  const long big_tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  if (fast && force_generic != 2 && (force_generic == 3 || (big_tiles >= 192 && M >= 256 && N >= 256))) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_256<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute((const void*)gemm_nt_256<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      attr_done = true;
    }
    if (out_f32) hipLaunchKernelGGL(gemm_nt_256<true>, dim3((unsigned)big_tiles), dim3(512), 131072, stream, p);
    else hipLaunchKernelGGL(gemm_nt_256<false>, dim3((unsigned)big_tiles), dim3(512), 131072, stream, p);
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

int vp_transpose_bf16(int rows, int cols, const void* in, long ld_in, void* out, long ld_out, hipStream_t stream) {
  VP_REQUIRE(rows > 0 && cols > 0 && in && out, VP_ERR_BAD_ARG, "vp_transpose_bf16: bad args");
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, rows, cols,
                     ld_in, ld_out);
  return vp_check_launch("vp_transpose_bf16");
}

}  // extern "C"
