// NHWC convolution helpers for the frozen DPT depth decoder (reference: ola_vlm/model/aux_heads/da_v2_head.py:182-321,
// called under no_grad at base_ola_vlm.py:462-470).  Every convolution of that decoder runs as the bf16 MFMA GEMM of gemm.hip:
//   3x3 (stride 1 / 2, pad 1)  -> im2col3x3 (optionally with the ResidualConvUnit's input ReLU) + GEMM [rows, 9C] x [Cout, 9C]^T
//   1x1                        -> the GEMM itself (NHWC rows are already [pixels, C])
//   ConvTranspose2d(k = s)     -> GEMM to [pixels, k*k*Cout] + pixel_shuffle
// plus the align_corners=True bilinear resize and the per-image min-max normalisation.  All HBM-bound: 16-byte vector I/O.
#include "common.h"

// col[(b, oy, ox), (ky*3 + kx)*C + c] = relu?(x[b, oy*s + ky - 1, ox*s + kx - 1, c])   (zero padding)
__global__ void im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int B, int H, int W, int C, int stride,
                                 int Ho, int Wo, int relu_in) {
  const int cv = C >> 3;
  const long total = (long)B * Ho * Wo * 9 * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int tap = (int)(r % 9);
    r /= 9;
    const int ox = (int)(r % Wo);
    const int oy = (int)((r / Wo) % Ho);
    const int b = (int)(r / ((long)Wo * Ho));
    const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      v = *(const bf16x8*)(x + (((long)b * H + iy) * W + ix) * C + c8 * 8);
      if (relu_in) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] < 0) ? (short)0 : v[e];     // bf16 sign bit == int16 sign bit
      }
    }
    *(bf16x8*)(col + r * 9L * C + (long)tap * C + c8 * 8) = v;
  }
}

// torch upsample_bilinear2d, align_corners=True (aten UpSample.h area_pixel_compute_source_index / compute_scales_value):
// scale = (in - 1) / (out - 1) in fp32, src = scale * dst, i0 = (int)src, i1 = i0 + (i0 < in - 1), lambda1 = src - i0;
// value = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11) in fp32, rounded once to bf16.
// align_corners=False (aten area_pixel_compute_source_index): scale = in / out, src = max(scale * (dst + 0.5) - 0.5, 0).
__global__ void bilinear_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C, int Ho,
                                     int Wo, int align) {
  const int cv = C >> 3;
  const float sh = align ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
  const float sw = align ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
  const long total = (long)B * Ho * Wo * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int ox = (int)(r % Wo);
    const int oy = (int)((r / Wo) % Ho);
    const int b = (int)(r / ((long)Wo * Ho));
    const float fy = align ? sh * oy : fmaxf(sh * (oy + 0.5f) - 0.5f, 0.f), fx = align ? sw * ox : fmaxf(sw * (ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float h1 = fy - y0, h0 = 1.f - h1, w1 = fx - x0, w0 = 1.f - w1;
    const bf16_t* base = x + (long)b * H * W * C + c8 * 8;
    const bf16x8 v00 = *(const bf16x8*)(base + ((long)y0 * W + x0) * C), v01 = *(const bf16x8*)(base + ((long)y0 * W + x1) * C);
    const bf16x8 v10 = *(const bf16x8*)(base + ((long)y1 * W + x0) * C), v11 = *(const bf16x8*)(base + ((long)y1 * W + x1) * C);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = (short)f2bf(h0 * (w0 * bf2f((bf16_t)v00[e]) + w1 * bf2f((bf16_t)v01[e])) +
                         h1 * (w0 * bf2f((bf16_t)v10[e]) + w1 * bf2f((bf16_t)v11[e])));
    *(bf16x8*)(y + r * C + c8 * 8) = o;
  }
}

// in: [B*H*W, k*k*C] with column (ky*k + kx)*C + c  ->  out: [B, H*k, W*k, C]
__global__ void pixel_shuffle_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int H, int W, int k, int C) {
  const int cv = C >> 3;
  const long total = (long)B * H * W * k * k * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int kk = (int)(r % (k * k));
    r /= (k * k);
    const int xx = (int)(r % W);
    const int yy = (int)((r / W) % H);
    const int b = (int)(r / ((long)W * H));
    const bf16x8 v = *(const bf16x8*)(in + (r * (k * k) + kk) * (long)C + c8 * 8);
    const int oy = yy * k + kk / k, ox = xx * k + kk % k;
    *(bf16x8*)(out + (((long)b * H * k + oy) * (W * k) + ox) * C + c8 * 8) = v;
  }
}

// y[b, :] = (x[b, :] - min_b) / (max_b - min_b) with bf16 rounding after each op (base_ola_vlm.py:466-468 on bf16 tensors)
__global__ __launch_bounds__(1024) void minmax_norm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n) {
  __shared__ float smn[16], smx[16];
  const bf16_t* xb = x + blockIdx.x * n;
  float mn = INFINITY, mx = -INFINITY;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = bf2f(xb[i]);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = smn[0]; mx = smx[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); }
  const float den = bfround(mx - mn);
  bf16_t* yb = y + blockIdx.x * n;
  for (long i = threadIdx.x; i < n; i += blockDim.x) yb[i] = f2bf(bfround(bf2f(xb[i]) - mn) / den);
}

#define GRID_FOR(n) dim3((unsigned)min((long)8192, ((long)(n) + 255) / 256))

extern "C" {

int vp_im2col3x3_nhwc(int B, int H, int W, int C, int stride, int relu_in, const void* x, void* col, hipStream_t s) {
  VP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (stride == 1 || stride == 2), VP_ERR_BAD_ARG, "vp_im2col3x3_nhwc: bad args");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(im2col3x3_kernel, GRID_FOR((long)B * Ho * Wo * 9 * (C / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)col, B, H, W, C,
                     stride, Ho, Wo, relu_in);
  return vp_check_launch("vp_im2col3x3_nhwc");
}

int vp_bilinear_nhwc(int B, int H, int W, int C, int Ho, int Wo, int align_corners, const void* x, void* y, hipStream_t s) {
  VP_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, VP_ERR_BAD_ARG, "vp_bilinear_nhwc: bad args");
  hipLaunchKernelGGL(bilinear_nhwc_kernel, GRID_FOR((long)B * Ho * Wo * (C / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo, align_corners);
  return vp_check_launch("vp_bilinear_nhwc");
}

int vp_pixel_shuffle_nhwc(int B, int H, int W, int k, int C, const void* x, void* y, hipStream_t s) {
  VP_REQUIRE(B > 0 && H > 0 && W > 0 && k > 0 && C > 0 && C % 8 == 0, VP_ERR_BAD_ARG, "vp_pixel_shuffle_nhwc: bad args");
  hipLaunchKernelGGL(pixel_shuffle_kernel, GRID_FOR((long)B * H * W * k * k * (C / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, k, C);
  return vp_check_launch("vp_pixel_shuffle_nhwc");
}

int vp_minmax_norm(int B, long n, const void* x, void* y, hipStream_t s) {
  VP_REQUIRE(B > 0 && n > 0, VP_ERR_BAD_ARG, "vp_minmax_norm: bad args");
  hipLaunchKernelGGL(minmax_norm_kernel, dim3(B), dim3(1024), 0, s, (const bf16_t*)x, (bf16_t*)y, n);
  return vp_check_launch("vp_minmax_norm");
}

}  // extern "C"
