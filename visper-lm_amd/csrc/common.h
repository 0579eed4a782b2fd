// Shared device helpers for the VisPer-LM MI355X (gfx950 / CDNA4) hot-path kernels.
// wave = 64 lanes; MFMA 16x16x32 bf16 fragments: A/B = 8 bf16 per lane, C/D = 4 fp32 per lane
// (C/D: col = lane & 15, row = (lane >> 4) * 4 + reg).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include <type_traits>

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) — the index is a constant expression inside f (asm "n" operands,
// if constexpr), which a `#pragma unroll` loop variable is only after the optimiser ran (and only if it did unroll)
template <class F, int... I>
__device__ __forceinline__ void vp_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void vp_static_for(F&& f) { vp_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

typedef uint16_t bf16_t;                                             // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) short bf16x8;            // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;             // MFMA C/D fragment
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define VP_OK 0
#define VP_ERR_BAD_ARG (-1)
#define VP_ERR_UNSUPPORTED_SHAPE (-2)
#define VP_ERR_HIP (-3)

void vp_set_error(const char* fmt, ...);
int vp_check_launch(const char* what);

#define VP_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      vp_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, ((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }   // RNE (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// two floats -> packed bf16 pair (lo | hi << 16), RNE: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64).  `red` = >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }

// XCD-aware bijective block remap (8 XCDs, dispatcher places block b on XCD b % 8): gives each
// XCD a contiguous range of logical tile ids so neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  const int xcd = bid % nx, idx = bid / nx;
  const int q = nwg / nx, r = nwg % nx;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
