// RMSNorm / LayerNorm forward + backward (HBM-bound; one wave per row, 16-byte bf16 loads,
// row kept in registers between the statistics pass and the normalise pass, fp32 math).
#include "common.h"

constexpr int MAX_NCH = 16;   // up to 16 chunks of 8 bf16 per lane  ->  H <= 8192 on the register path
// kernels are templated on NCH = ceil(H / 512) in {2,4,8,16} so small rows do not pay the register cost

template <int NCH>
__device__ __forceinline__ void load_row(const bf16_t* x, int H, int lane, bf16x8 (&v)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) v[c] = *(const bf16x8*)(x + e);
  }
}

// ---------------------------------------------------------------- RMSNorm ----------------------
// HF LlamaRMSNorm: y = w * bf16(x * rsqrt(mean(x^2) + eps))   (two bf16 roundings)
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out, int M,
                                                          int H, long ldx, long ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16_t* xr = x + (long)row * ldx;
  bf16x8 v[NCH];
  load_row<NCH>(xr, H, lane, v);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if ((c * 64 + lane) * 8 < H)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = bf2f((bf16_t)v[c][j]); ss += f * f; }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
  bf16_t* yr = y + (long)row * ldy;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      const bf16x8 wv = *(const bf16x8*)(w + e);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = (short)f2bf(bf2f((bf16_t)wv[j]) * bfround(bf2f((bf16_t)v[c][j]) * rstd));
      *(bf16x8*)(yr + e) = o;
    }
  }
}

// dx = rstd * (g - x * rstd^2 * mean(g*x)),  g = w*dy ;  optional  dx += dres  (residual-stream grad)
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, int M,
                                                          int H, long ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  bf16x8 xv[NCH], gv[NCH], rv[NCH];
#ifndef VP_NORM_NT
#define VP_NORM_NT 1
#endif
#if VP_NORM_NT >= 1
  // NON-TEMPORAL loads (round 5): the three streams are read once; 114.6 -> 102.7 us at [16384, 4096] over rotating buffers (4.68 -> 5.23 TB/s)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      xv[c] = __builtin_nontemporal_load((const bf16x8*)(x + (long)row * ld + e));
      gv[c] = __builtin_nontemporal_load((const bf16x8*)(dy + (long)row * ld + e));
      if (dres) rv[c] = __builtin_nontemporal_load((const bf16x8*)(dres + (long)row * ld + e));
    }
  }
#else
  load_row<NCH>(x + (long)row * ld, H, lane, xv);
  load_row<NCH>(dy + (long)row * ld, H, lane, gv);
  if (dres) load_row<NCH>(dres + (long)row * ld, H, lane, rv);      // all three streams in flight before the row reduction (was: after it)
#endif
  const float rstd = rstd_in[row];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      const bf16x8 wv = *(const bf16x8*)(w + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += bf2f((bf16_t)wv[j]) * bf2f((bf16_t)gv[c][j]) * bf2f((bf16_t)xv[c][j]);
    }
  }
  dot = wave_sum(dot);
  const float coef = dot * rstd * rstd / (float)H;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      const bf16x8 wv = *(const bf16x8*)(w + e);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = bf2f((bf16_t)wv[j]) * bf2f((bf16_t)gv[c][j]);
        float d = rstd * (g - bf2f((bf16_t)xv[c][j]) * coef);
        if (dres) d += bf2f((bf16_t)rv[c][j]);
        o[j] = (short)f2bf(d);
      }
#if defined(VP_NORM_NT) && VP_NORM_NT >= 2
      __builtin_nontemporal_store(o, (bf16x8*)(dx + (long)row * ld + e));
#else
      *(bf16x8*)(dx + (long)row * ld + e) = o;
#endif
    }
  }
}

// one wave per row: lanes stride over the row's partials (float4 each), then the fixed xor tree of wave_sum: deterministic
__global__ __launch_bounds__(256) void rstd_from_sumsq_kernel(const float* __restrict__ part, float* __restrict__ rstd, int M, int nparts,
                                                              float inv_h, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const f32x4* p4 = (const f32x4*)(part + (long)row * nparts);
  float s = 0.f;
  for (int j = lane; j < (nparts >> 2); j += 64) {
    const f32x4 v = p4[j];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  s = wave_sum(s);
  if (lane == 0) rstd[row] = rsqrtf(s * inv_h + eps);
}

// ---------------------------------------------------------------- LayerNorm --------------------
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int M, int H, long ldx, long ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  bf16x8 v[NCH];
  load_row<NCH>(x + (long)row * ldx, H, lane, v);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if ((c * 64 + lane) * 8 < H)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += bf2f((bf16_t)v[c][j]);
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if ((c * 64 + lane) * 8 < H)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = bf2f((bf16_t)v[c][j]) - mean; ss += d * d; }
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      const bf16x8 wv = *(const bf16x8*)(w + e);
      const bf16x8 bv = *(const bf16x8*)(b + e);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = (short)f2bf((bf2f((bf16_t)v[c][j]) - mean) * rstd * bf2f((bf16_t)wv[j]) + bf2f((bf16_t)bv[j]));
      *(bf16x8*)(y + (long)row * ldy + e) = o;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*w ; optional dx += dres
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w, const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in,
                                                               const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, int M,
                                                               int H, long ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  bf16x8 xv[NCH], gv[NCH];
  load_row<NCH>(x + (long)row * ld, H, lane, xv);
  load_row<NCH>(dy + (long)row * ld, H, lane, gv);
  const float mean = mean_in[row], rstd = rstd_in[row];
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      const bf16x8 wv = *(const bf16x8*)(w + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = bf2f((bf16_t)wv[j]) * bf2f((bf16_t)gv[c][j]);
        sg += g;
        sgx += g * (bf2f((bf16_t)xv[c][j]) - mean) * rstd;
      }
    }
  }
  sg = wave_sum(sg) / (float)H;
  sgx = wave_sum(sgx) / (float)H;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < H) {
      const bf16x8 wv = *(const bf16x8*)(w + e);
      bf16x8 rv;
      if (dres) rv = *(const bf16x8*)(dres + (long)row * ld + e);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = bf2f((bf16_t)wv[j]) * bf2f((bf16_t)gv[c][j]);
        const float xh = (bf2f((bf16_t)xv[c][j]) - mean) * rstd;
        float d = rstd * (g - sg - xh * sgx);
        if (dres) d += bf2f((bf16_t)rv[j]);
        o[j] = (short)f2bf(d);
      }
      *(bf16x8*)(dx + (long)row * ld + e) = o;
    }
  }
}

// dw[c] = sum_rows dy*xhat, db[c] = sum_rows dy : stage 1 = per-row-slab partials (deterministic),
// thread t owns columns t, t+256, ... ; stage 2 (colsum_finish in elementwise.hip) sums the slabs.
__global__ __launch_bounds__(256) void layernorm_bwd_wb_partial_kernel(const bf16_t* __restrict__ dy,
                                                                       const bf16_t* __restrict__ x,
                                                                       const float* __restrict__ mean_in,
                                                                       const float* __restrict__ rstd_in,
                                                                       float* __restrict__ pw, float* __restrict__ pb, int M,
                                                                       int H, long ld, int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int c = blockIdx.y * 256 + threadIdx.x; c < H; c += gridDim.y * 256) {
    float aw = 0.f, ab = 0.f;
    for (int r = r0; r < r1; ++r) {
      const float d = bf2f(dy[(long)r * ld + c]);
      aw += d * (bf2f(x[(long)r * ld + c]) - mean_in[r]) * rstd_in[r];
      ab += d;
    }
    pw[(long)blockIdx.x * H + c] = aw;
    pb[(long)blockIdx.x * H + c] = ab;
  }
}

#define NORM_DISPATCH(KERNEL, H, ...)                                                                   \
  do {                                                                                                \
    const int nch_ = ((H) + 511) / 512;                                                               \
    if (nch_ <= 2) hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__);                                       \
    else if (nch_ <= 4) hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__);                                  \
    else if (nch_ <= 8) hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__);                                  \
    else hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__);                                                \
  } while (0)

extern "C" {

static int check_norm(const char* what, int M, int H) {
  VP_REQUIRE(M > 0 && H > 0, VP_ERR_BAD_ARG, "%s: bad dims", what);
  VP_REQUIRE(H % 8 == 0 && H <= MAX_NCH * 512, VP_ERR_UNSUPPORTED_SHAPE, "%s: H=%d must be a multiple of 8 and <= %d", what, H,
             MAX_NCH * 512);
  return VP_OK;
}

int vp_rmsnorm_fwd(int M, int H, const void* x, long ldx, const void* w, float eps, void* y, long ldy, float* rstd,
                   hipStream_t s) {
  int e = check_norm("vp_rmsnorm_fwd", M, H);
  if (e) return e;
  NORM_DISPATCH(rmsnorm_fwd_kernel, H, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y,
                     rstd, M, H, ldx, ldy, eps);
  return vp_check_launch("vp_rmsnorm_fwd");
}

int vp_rmsnorm_bwd(int M, int H, const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                   long ld, hipStream_t s) {
  int e = check_norm("vp_rmsnorm_bwd", M, H);
  if (e) return e;
  NORM_DISPATCH(rmsnorm_bwd_kernel, H, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, M, H, ld);
  return vp_check_launch("vp_rmsnorm_bwd");
}

// rstd[row] = rsqrt(sum_j part[row, j] / H + eps): finishes the per-16-column sums of squares vp_gemm_bf16_sumsq emitted (nparts = N / 16), in a
// fixed order (deterministic).  Reference: HF LlamaRMSNorm.forward (variance = hidden.pow(2).mean(-1); rsqrt(variance + eps)).
int vp_rstd_from_sumsq(int M, int nparts, const float* part, int H, float eps, float* rstd, hipStream_t s) {
  VP_REQUIRE(M > 0 && nparts > 0 && nparts % 4 == 0 && H > 0 && part && rstd, VP_ERR_BAD_ARG, "vp_rstd_from_sumsq: bad args");
  hipLaunchKernelGGL(rstd_from_sumsq_kernel, dim3((M + 3) / 4), dim3(256), 0, s, part, rstd, M, nparts, 1.f / (float)H, eps);
  return vp_check_launch("vp_rstd_from_sumsq");
}

int vp_layernorm_fwd(int M, int H, const void* x, long ldx, const void* w, const void* b, float eps, void* y, long ldy,
                     float* mean, float* rstd, hipStream_t s) {
  int e = check_norm("vp_layernorm_fwd", M, H);
  if (e) return e;
  NORM_DISPATCH(layernorm_fwd_kernel, H, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)b, (bf16_t*)y, mean, rstd, M, H, ldx, ldy, eps);
  return vp_check_launch("vp_layernorm_fwd");
}

int vp_layernorm_bwd_dx(int M, int H, const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                        const void* dres, void* dx, long ld, hipStream_t s) {
  int e = check_norm("vp_layernorm_bwd_dx", M, H);
  if (e) return e;
  NORM_DISPATCH(layernorm_bwd_dx_kernel, H, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)w, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, M, H, ld);
  return vp_check_launch("vp_layernorm_bwd_dx");
}

// partial buffers pw/pb: [n_slabs, H] fp32 with n_slabs = ceil(M / rows_per_block); finish with vp_colsum_finish.
int vp_layernorm_bwd_wb_partial(int M, int H, const void* dy, const void* x, const float* mean, const float* rstd, float* pw,
                                float* pb, long ld, int rows_per_block, hipStream_t s) {
  VP_REQUIRE(M > 0 && H > 0 && rows_per_block > 0, VP_ERR_BAD_ARG, "vp_layernorm_bwd_wb_partial: bad args");
  dim3 grid((M + rows_per_block - 1) / rows_per_block, (H + 255) / 256);
  hipLaunchKernelGGL(layernorm_bwd_wb_partial_kernel, grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd,
                     pw, pb, M, H, ld, rows_per_block);
  return vp_check_launch("vp_layernorm_bwd_wb_partial");
}

}  // extern "C"
