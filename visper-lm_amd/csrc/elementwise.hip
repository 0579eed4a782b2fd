// HBM-bound elementwise / gather / reduction kernels of the PT train step (all 16-byte vectorised,
// grid-stride, fp32 math, bf16 I/O): RoPE, SwiGLU fwd/bwd, GELU/ReLU fwd/bwd, residual add, column sums
// (bias grads), row gather / gather-sum (sequence splice fwd/bwd), casts, fused AdamW.
#include "common.h"

#define GRID_FOR(n) dim3((unsigned)min((long)2048, ((long)(n) + 255) / 256))

// ---------------------------------------------------------------- RoPE (rotate-half) -----------
// x: [T tokens, nheads*hd] slice with token stride ld; cos/sin: fp32 [S, hd/2] (already rounded to the
// activation dtype by the host, as HF does); pos[t] = position id of token t (NULL: t % S).
// inverse=1 applies the transpose rotation (backward).
__global__ void rope_kernel(bf16_t* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                            const int* __restrict__ pos, long T, int S, int nheads, int hd, long ld, int inverse) {
  const int half = hd >> 1;
  const int vec_per_head = half >> 3;                   // 8-wide vectors in one half
  const long total = T * nheads * vec_per_head;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vec_per_head);
    const int h = (int)((i / vec_per_head) % nheads);
    const long t = i / ((long)vec_per_head * nheads);
    const int p = pos ? pos[t] : (int)(t % S);
    bf16_t* base = x + t * ld + (long)h * hd + v * 8;
    bf16x8 a = *(bf16x8*)base, b = *(bf16x8*)(base + half);
    const float* c = cs + (long)p * half + v * 8;
    const float* s = sn + (long)p * half + v * 8;
    bf16x8 oa, ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x1 = bf2f((bf16_t)a[j]), x2 = bf2f((bf16_t)b[j]);
      const float cc = c[j], ss = inverse ? -s[j] : s[j];
      // HF: q*cos + rotate_half(q)*sin, each product rounded to bf16 before the add
      oa[j] = (short)f2bf(bfround(x1 * cc) + bfround(-x2 * ss));
      ob[j] = (short)f2bf(bfround(x2 * cc) + bfround(x1 * ss));
    }
    *(bf16x8*)base = oa;
    *(bf16x8*)(base + half) = ob;
  }
}

// ---------------------------------------------------------------- SwiGLU -----------------------
// gu: [M, 2F] with gate / up interleaved in 8-wide chunks (g0..7 | u0..7 | g8..15 | ...; the layout the fused GEMM
// epilogues produce and consume);  out[M,F] = silu(gate) * up
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, long M, int F, long ldg, long ldo, int il) {
  const int vpr = F >> 3;
  const long total = M * vpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vpr;
    const int c = (int)(i % vpr) * 8;
    const long go = il ? 2L * c : c, uo = il ? 2L * c + 8 : (long)F + c;      // interleaved 8-chunks | [gate | up] halves
    const bf16x8 g = *(const bf16x8*)(gu + r * ldg + go), u = *(const bf16x8*)(gu + r * ldg + uo);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bfround(silu(bf2f((bf16_t)g[j]))) * bf2f((bf16_t)u[j]));
    *(bf16x8*)(out + r * ldo + c) = o;
  }
}

// dgu[M,2F]: dgate = dact * up * silu'(gate), dup = dact * silu(gate)
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                  long M, int F, long ldd, long ldg, int il) {
  const int vpr = F >> 3;
  const long total = M * vpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vpr;
    const int c = (int)(i % vpr) * 8;
    const long go = il ? 2L * c : c, uo = il ? 2L * c + 8 : (long)F + c;      // interleaved 8-chunks | [gate | up] halves
    const bf16x8 g = *(const bf16x8*)(gu + r * ldg + go), u = *(const bf16x8*)(gu + r * ldg + uo);
    const bf16x8 d = *(const bf16x8*)(dact + r * ldd + c);
    bf16x8 og, ou;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gg = bf2f((bf16_t)g[j]), uu = bf2f((bf16_t)u[j]), dd = bf2f((bf16_t)d[j]);
      const float sg = 1.f / (1.f + __expf(-gg));
      og[j] = (short)f2bf(dd * uu * sg * (1.f + gg * (1.f - sg)));
      ou[j] = (short)f2bf(dd * gg * sg);
    }
    *(bf16x8*)(dgu + r * ldg + go) = og;
    *(bf16x8*)(dgu + r * ldg + uo) = ou;
  }
}

// ---------------------------------------------------------------- activations ------------------
// kind: 1 = GELU(erf), 3 = ReLU   (same enum as the GEMM epilogue)
__global__ void act_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n, int kind) {
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    const bf16x8 v = *(const bf16x8*)(x + i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf2f((bf16_t)v[j]);
      o[j] = (short)f2bf(kind == 1 ? gelu_erf(f) : fmaxf(f, 0.f));
    }
    *(bf16x8*)(y + i) = o;
  }
}
__global__ void act_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, long n,
                               int kind) {
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    const bf16x8 v = *(const bf16x8*)(x + i), d = *(const bf16x8*)(dy + i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf2f((bf16_t)v[j]);
      const float gr = kind == 1 ? gelu_erf_grad(f) : (f > 0.f ? 1.f : 0.f);
      o[j] = (short)f2bf(bf2f((bf16_t)d[j]) * gr);
    }
    *(bf16x8*)(dx + i) = o;
  }
}

// out = a + b (bf16), n % 8 == 0
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long n) {
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    const bf16x8 x = *(const bf16x8*)(a + i), y = *(const bf16x8*)(b + i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bf2f((bf16_t)x[j]) + bf2f((bf16_t)y[j]));
    *(bf16x8*)(out + i) = o;
  }
}

// strided 2-D add: dst[r, c] += src[r, c]   (bf16, C % 8 == 0)
__global__ void add2d_kernel(bf16_t* __restrict__ dst, long ldd, const bf16_t* __restrict__ src, long lds, long R, int C) {
  const int vpr = C >> 3;
  const long total = R * vpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vpr;
    const int c = (int)(i % vpr) * 8;
    const bf16x8 x = *(const bf16x8*)(dst + r * ldd + c), y = *(const bf16x8*)(src + r * lds + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bf2f((bf16_t)x[j]) + bf2f((bf16_t)y[j]));
    *(bf16x8*)(dst + r * ldd + c) = o;
  }
}

// strided 2-D copy (bf16, C % 8 == 0)
__global__ void copy2d_kernel(bf16_t* __restrict__ dst, long ldd, const bf16_t* __restrict__ src, long lds, long R, int C) {
  const int vpr = C >> 3;
  const long total = R * vpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vpr;
    const int c = (int)(i % vpr) * 8;
    *(bf16x8*)(dst + r * ldd + c) = *(const bf16x8*)(src + r * lds + c);
  }
}

// ---------------------------------------------------------------- column sums (bias grads) -----
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long M,
                                                             int N, long ld, int rows_per_block) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  for (int c = blockIdx.y * 256 + threadIdx.x; c < N; c += gridDim.y * 256) {
    float a = 0.f;
    for (long r = r0; r < r1; ++r) a += bf2f(x[r * ld + c]);
    part[(long)blockIdx.x * N + c] = a;
  }
}
// out[c] (+)= scale * sum_s part[s, c]
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nslab, int N,
                                                            float scale, int accumulate) {
  // 64 columns per block, the slabs split over the block's 4 waves with 4 independent partial sums each (the old one-thread-per-column
  // serial loop over up to 256 slabs took 45 us; this is launch-bound)
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < N) {
    int s = sg;
    for (; s + 12 < nslab; s += 16) {
      a0 += part[(long)s * N + c];
      a1 += part[(long)(s + 4) * N + c];
      a2 += part[(long)(s + 8) * N + c];
      a3 += part[(long)(s + 12) * N + c];
    }
    for (; s < nslab; s += 4) a0 += part[(long)s * N + c];
  }
  red[sg][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sg == 0 && c < N) {
    const float a = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * scale;
    out[c] = accumulate ? out[c] + a : a;
  }
}

// ---------------------------------------------------------------- row gather (splice fwd) ------
// out[i,:] = srcs[kind[i]][row[i],:]   (kind < 0 -> zeros).  H % 8 == 0.  One wave per output row.
struct GatherSrcs {
  const bf16_t* p[4];
  long ld[4];
};
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherSrcs s, const int* __restrict__ kind, const int* __restrict__ row,
                                                          bf16_t* __restrict__ out, long ldo, long n_out, int H) {
  const int lane = threadIdx.x & 63;
  for (long i = blockIdx.x * 4L + (threadIdx.x >> 6); i < n_out; i += gridDim.x * 4L) {
    const int k = kind[i];
    const bf16_t* src = k >= 0 ? s.p[k] + (long)row[i] * s.ld[k] : nullptr;
    for (int e = lane * 8; e < H; e += 512) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (src) v = *(const bf16x8*)(src + e);
      *(bf16x8*)(out + i * ldo + e) = v;
    }
  }
}

// out[i,:] = scale * sum_{k<cnt} src[idx[i*cnt+k],:]  (idx < 0 skipped); fp32 accumulate; out bf16 or f32.
template <bool OUT_F32, bool SRC_F32>
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const void* __restrict__ src_, long lds, const int* __restrict__ idx,
                                                              int cnt, float scale, void* __restrict__ out_, long ldo, long n_out,
                                                              int H, int accumulate) {
  for (long i = blockIdx.y; i < n_out; i += gridDim.y) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < H; c += gridDim.x * 256) {
      float a = 0.f;
      for (int k = 0; k < cnt; ++k) {
        const int r = idx[i * cnt + k];
        if (r >= 0) a += SRC_F32 ? ((const float*)src_)[(long)r * lds + c] : bf2f(((const bf16_t*)src_)[(long)r * lds + c]);
      }
      a *= scale;
      if (OUT_F32) {
        float* o = (float*)out_ + i * ldo + c;
        *o = accumulate ? *o + a : a;
      } else {
        ((bf16_t*)out_)[i * ldo + c] = f2bf(a);
      }
    }
  }
}

// ---------------------------------------------------------------- casts / fills / reductions ---
__global__ void cast_f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ void cast_bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n, int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = accumulate ? y[i] + bf2f(x[i]) : bf2f(x[i]);
}
// dst[idx[r], :] = float(src[r, :])  (idx NULL: r; idx < 0 skips the row): bf16 logits of a row chunk widened into their rows of the fp32
// [B*S, V] tensor the reference's forward returns (ola_llama.py:121-122 `logits.float()`).  One wave per row, 16-byte loads, write-once
// (non-temporal) 2 x 16-byte stores.  H % 8 == 0.
__global__ __launch_bounds__(256) void scatter_rows_bf16_to_f32_kernel(const bf16_t* __restrict__ src, long lds, const int* __restrict__ idx,
                                                                       float* __restrict__ dst, long ldd, long n, int H) {
  const int lane = threadIdx.x & 63;
  for (long r = blockIdx.x * 4L + (threadIdx.x >> 6); r < n; r += gridDim.x * 4L) {
    const long t = idx ? (long)idx[r] : r;
    if (t < 0) continue;
    const bf16_t* s = src + r * lds;
    float* d = dst + t * ldd;
    for (int e = lane * 8; e < H; e += 512) {
      const bf16x8 v = __builtin_nontemporal_load((const bf16x8*)(s + e));
      const f32x4 lo = {bf2f((bf16_t)v[0]), bf2f((bf16_t)v[1]), bf2f((bf16_t)v[2]), bf2f((bf16_t)v[3])};
      const f32x4 hi = {bf2f((bf16_t)v[4]), bf2f((bf16_t)v[5]), bf2f((bf16_t)v[6]), bf2f((bf16_t)v[7])};
      __builtin_nontemporal_store(lo, (f32x4*)(d + e));
      __builtin_nontemporal_store(hi, (f32x4*)(d + e + 4));
    }
  }
}

// deterministic sum of n floats -> out[0] (single block; n is small: per-row losses / partials)
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long n, float scale) {
  __shared__ float red[16];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) a += x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a * scale;
}

// dst[idx[r], :] += src[r, :] (fp32 atomics; idx < 0 skips the row): embedding-table gradient of a trainable LLM (IFT stage).
// Like torch's index_add_ / embedding backward on GPUs the accumulation order is not deterministic.
__global__ void scatter_add_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, long n,
                                        int H, long lds) {
  const int cv = H >> 3;
  const long total = n * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cv;
    const int c = (int)(i % cv) * 8;
    const int t = idx[r];
    if (t < 0) continue;
    const bf16x8 v = *(const bf16x8*)(src + r * lds + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(dst + (long)t * H + c + e, bf2f((bf16_t)v[e]));
  }
}

// sum of squares of n floats (gradient-clipping norm): per-block partials, then sum_f32_kernel over them (deterministic)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long n) {
  __shared__ float red[16];
  float a = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) a += x[i] * x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// ---------------------------------------------------------------- fused AdamW ------------------
// fp32 master params/grads/moments (flat), optional bf16 shadow written in the same pass.
// torch.optim.AdamW semantics: p *= 1 - lr*wd ; m,v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16_t* __restrict__ shadow, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2_sqrt, float gscale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = f2bf(pi);
  }
}
// four parameters per thread and non-temporal streams (round 5): every buffer is read and written exactly once per step (the IFT stage moves 30 bytes
// for each of 8 G parameters).  Same arithmetic per element as adamw_kernel: bit-identical.  n4 = n / 4 elements of 16-byte aligned buffers.
__global__ void adamw4_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v,
                              bf16x4* __restrict__ shadow, long n4, float lr, float b1, float b2, float eps, float wd, float bc1,
                              float bc2_sqrt, float gscale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g4 = __builtin_nontemporal_load(g + i), m4 = __builtin_nontemporal_load(m + i), v4 = __builtin_nontemporal_load(v + i);
    f32x4 p4 = __builtin_nontemporal_load(p + i), mo, vo;
    bf16x4 s4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = g4[e] * gscale;
      float pi = p4[e] * (1.f - lr * wd);
      const float mi = b1 * m4[e] + (1.f - b1) * gr;
      const float vi = b2 * v4[e] + (1.f - b2) * gr * gr;
      pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
      p4[e] = pi; mo[e] = mi; vo[e] = vi;
      s4[e] = (short)f2bf(pi);
    }
    __builtin_nontemporal_store(p4, p + i);
    __builtin_nontemporal_store(mo, m + i);
    __builtin_nontemporal_store(vo, v + i);
    if (shadow) shadow[i] = s4;                        // (read by the next step's GEMMs: ordinary store)
  }
}


// ---------------------------------------------------------------- depthwise 7x7 conv (ConvNeXt block, NHWC) ----------
// x, y: [B, Hh, Ww, C] bf16 (channels last); w: [49, C] bf16 (tap-major, so 8 channels of one tap are one 16-byte load);
// bias: [C].  Zero padding 3; fp32 accumulate (bias first, then taps in (dy, dx) order), one bf16 rounding (torch conv2d).
// timm ConvNeXtBlock.conv_dw (clip_convnext_encoder.py:161-165).
// One thread = 8 channels of PX CONSECUTIVE output pixels of a row: an input vector is loaded and converted once and feeds up to 7 outputs,
// the 7 taps of a filter row are converted once per PX outputs, the multiply-adds are packed (v_pk_fma_f32, two channels each).  Round 4:
// replaces the one-pixel-per-thread kernel (784 loads, ~10 400 VALU per 64 outputs: 602 / 283 / 147 / 76 us at the four ConvNeXt-XXL stage
// shapes of configs[3]); PX = 4 measures 405 / 176 / 90 / 48 us (PX = 8: 413 / 194 / 99 / 54), about half L1 traffic (17 16-byte loads per
// filter row and thread) and half VALU issue.  Same accumulation order as before (bias, then taps in (dy, dx) order, fused multiply-adds).
template <int PX>
__global__ __launch_bounds__(256) void dwconv7x7_nhwc_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int B, int Hh,
                                                             int Ww, int C) {
  const int cv = C >> 3, nxt = (Ww + PX - 1) / PX;
  const long total = (long)B * Hh * nxt * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv) * 8;
    long t = i / cv;
    const int x0 = (int)(t % nxt) * PX;
    t /= nxt;
    const int yh = (int)(t % Hh);
    const long b = t / Hh;
    f32x2 acc[PX][4];
    {
      const bf16x8 bv = *(const bf16x8*)(bias + c8);
#pragma unroll
      for (int px = 0; px < PX; ++px)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[px][q] = f32x2{bf2f((bf16_t)bv[2 * q]), bf2f((bf16_t)bv[2 * q + 1])};
    }
    for (int dy = 0; dy < 7; ++dy) {
      const int yy = yh + dy - 3;
      if (yy < 0 || yy >= Hh) continue;
      f32x2 wf[7][4];
#pragma unroll
      for (int dx = 0; dx < 7; ++dx) {
        const bf16x8 wv = *(const bf16x8*)(w + (dy * 7 + dx) * C + c8);
#pragma unroll
        for (int q = 0; q < 4; ++q) wf[dx][q] = f32x2{bf2f((bf16_t)wv[2 * q]), bf2f((bf16_t)wv[2 * q + 1])};
      }
      // the strip's PX + 6 input vectors as ONE batch of unconditional loads (columns clamped into the row, out-of-range ones zeroed after):
      // a bounds branch around each load left one L2 round trip exposed per tap (19 TFLOP/s; instruction-issue 25 % busy)
      const bf16_t* row = x + ((b * Hh + yy) * Ww) * C + c8;
      bf16x8 xv[PX + 6];
#pragma unroll
      for (int xi = 0; xi < PX + 6; ++xi) xv[xi] = *(const bf16x8*)(row + (long)min(max(x0 + xi - 3, 0), Ww - 1) * C);
#pragma unroll
      for (int xi = 0; xi < PX + 6; ++xi) {
        const int xx = x0 + xi - 3;
        const bool ok = xx >= 0 && xx < Ww;
        f32x2 xf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xf[q] = ok ? f32x2{bf2f((bf16_t)xv[xi][2 * q]), bf2f((bf16_t)xv[xi][2 * q + 1])} : f32x2{0.f, 0.f};
#pragma unroll
        for (int px = 0; px < PX; ++px) {
          const int dx = xi - px;
          if (dx >= 0 && dx < 7) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[px][q] = __builtin_elementwise_fma(xf[q], wf[dx][q], acc[px][q]);
          }
        }
      }
    }
    bf16_t* yrow = y + ((b * Hh + yh) * Ww) * C + c8;
#pragma unroll
    for (int px = 0; px < PX; ++px) {
      if (x0 + px < Ww) {
        const u32x4 o = {pack_bf16x2(acc[px][0][0], acc[px][0][1]), pack_bf16x2(acc[px][1][0], acc[px][1][1]),
                         pack_bf16x2(acc[px][2][0], acc[px][2][1]), pack_bf16x2(acc[px][3][0], acc[px][3][1])};
        *(u32x4*)(yrow + (long)(x0 + px) * C) = o;
      }
    }
  }
}

extern "C" {

int vp_dwconv7x7_nhwc(int B, int Hh, int Ww, int C, const void* x, const void* w, const void* bias, void* y, hipStream_t s) {
  VP_REQUIRE(B > 0 && Hh > 0 && Ww > 0 && C > 0 && C % 8 == 0 && x && w && bias && y, VP_ERR_BAD_ARG, "vp_dwconv7x7_nhwc: bad args");
  constexpr int DW_PX = 4;
  const long total = (long)B * Hh * ((Ww + DW_PX - 1) / DW_PX) * (C / 8);
  hipLaunchKernelGGL(dwconv7x7_nhwc_kernel<DW_PX>, dim3((unsigned)min(65536L, (total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)bias, (bf16_t*)y, B, Hh, Ww, C);
  return vp_check_launch("vp_dwconv7x7_nhwc");
}

int vp_rope(long T, int S, int nheads, int hd, void* x, long ld, const float* cos_t, const float* sin_t, const int* pos,
            int inverse, hipStream_t s) {
  VP_REQUIRE(T > 0 && nheads > 0 && hd > 0 && x && cos_t && sin_t, VP_ERR_BAD_ARG, "vp_rope: bad args");
  VP_REQUIRE(hd % 16 == 0 && ld % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE, "vp_rope: head_dim %d must be a multiple of 16", hd);
  const long total = T * nheads * (hd / 16);
  hipLaunchKernelGGL(rope_kernel, GRID_FOR(total), dim3(256), 0, s, (bf16_t*)x, cos_t, sin_t, pos, T, S, nheads, hd, ld, inverse);
  return vp_check_launch("vp_rope");
}

int vp_swiglu_fwd(long M, int F, const void* gate_up, long ldg, void* out, long ldo, int interleaved, hipStream_t s) {
  VP_REQUIRE(M > 0 && F > 0 && F % 8 == 0 && ldg % 8 == 0 && ldo % 8 == 0, VP_ERR_BAD_ARG, "vp_swiglu_fwd: bad args");
  hipLaunchKernelGGL(swiglu_fwd_kernel, GRID_FOR(M * (F / 8)), dim3(256), 0, s, (const bf16_t*)gate_up, (bf16_t*)out, M, F, ldg, ldo, interleaved);
  return vp_check_launch("vp_swiglu_fwd");
}

int vp_swiglu_bwd(long M, int F, const void* dact, long ldd, const void* gate_up, void* dgate_up, long ldg, int interleaved,
                  hipStream_t s) {
  VP_REQUIRE(M > 0 && F > 0 && F % 8 == 0 && ldg % 8 == 0 && ldd % 8 == 0, VP_ERR_BAD_ARG, "vp_swiglu_bwd: bad args");
  hipLaunchKernelGGL(swiglu_bwd_kernel, GRID_FOR(M * (F / 8)), dim3(256), 0, s, (const bf16_t*)dact, (const bf16_t*)gate_up,
                     (bf16_t*)dgate_up, M, F, ldd, ldg, interleaved);
  return vp_check_launch("vp_swiglu_bwd");
}

int vp_act_fwd(int kind, long n, const void* x, void* y, hipStream_t s) {
  VP_REQUIRE(n > 0 && n % 8 == 0 && (kind == 1 || kind == 3), VP_ERR_BAD_ARG, "vp_act_fwd: bad args");
  hipLaunchKernelGGL(act_fwd_kernel, GRID_FOR(n / 8), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n, kind);
  return vp_check_launch("vp_act_fwd");
}

int vp_act_bwd(int kind, long n, const void* dy, const void* x, void* dx, hipStream_t s) {
  VP_REQUIRE(n > 0 && n % 8 == 0 && (kind == 1 || kind == 3), VP_ERR_BAD_ARG, "vp_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, GRID_FOR(n / 8), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, n, kind);
  return vp_check_launch("vp_act_bwd");
}

int vp_add_bf16(long n, const void* a, const void* b, void* out, hipStream_t s) {
  VP_REQUIRE(n > 0 && n % 8 == 0, VP_ERR_BAD_ARG, "vp_add_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(add_kernel, GRID_FOR(n / 8), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
  return vp_check_launch("vp_add_bf16");
}

int vp_add2d_bf16(long R, int C, void* dst, long ldd, const void* src, long lds, hipStream_t s) {
  VP_REQUIRE(R > 0 && C > 0 && C % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0, VP_ERR_BAD_ARG, "vp_add2d_bf16: bad args");
  hipLaunchKernelGGL(add2d_kernel, GRID_FOR(R * (C / 8)), dim3(256), 0, s, (bf16_t*)dst, ldd, (const bf16_t*)src, lds, R, C);
  return vp_check_launch("vp_add2d_bf16");
}

int vp_copy2d_bf16(long R, int C, void* dst, long ldd, const void* src, long lds, hipStream_t s) {
  VP_REQUIRE(R > 0 && C > 0 && C % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0, VP_ERR_BAD_ARG, "vp_copy2d_bf16: bad args");
  hipLaunchKernelGGL(copy2d_kernel, GRID_FOR(R * (C / 8)), dim3(256), 0, s, (bf16_t*)dst, ldd, (const bf16_t*)src, lds, R, C);
  return vp_check_launch("vp_copy2d_bf16");
}

int vp_colsum_partial(long M, int N, const void* x, long ld, float* part, int rows_per_block, hipStream_t s) {
  VP_REQUIRE(M > 0 && N > 0 && rows_per_block > 0, VP_ERR_BAD_ARG, "vp_colsum_partial: bad args");
  dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block), (N + 255) / 256);
  hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, part, M, N, ld, rows_per_block);
  return vp_check_launch("vp_colsum_partial");
}

int vp_colsum_finish(int nslab, int N, const float* part, float* out, float scale, int accumulate, hipStream_t s) {
  VP_REQUIRE(nslab > 0 && N > 0, VP_ERR_BAD_ARG, "vp_colsum_finish: bad args");
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(256), 0, s, part, out, nslab, N, scale, accumulate);
  return vp_check_launch("vp_colsum_finish");
}

int vp_gather_rows(long n_out, int H, const void* const* srcs, const long* lds, int nsrc, const int* kind, const int* row, void* out,
                   long ldo, hipStream_t s) {
  VP_REQUIRE(n_out > 0 && H > 0 && H % 8 == 0 && nsrc >= 1 && nsrc <= 4, VP_ERR_BAD_ARG, "vp_gather_rows: bad args");
  GatherSrcs g{};
  for (int i = 0; i < nsrc; ++i) { g.p[i] = (const bf16_t*)srcs[i]; g.ld[i] = lds[i]; }
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)min(4096L, (n_out + 3) / 4)), dim3(256), 0, s, g, kind, row, (bf16_t*)out, ldo,
                     n_out, H);
  return vp_check_launch("vp_gather_rows");
}

int vp_gather_sum_rows(long n_out, int cnt, int H, const void* src, long lds, int src_f32, const int* idx, float scale, void* out,
                       long ldo, int out_f32, int accumulate, hipStream_t s) {
  VP_REQUIRE(n_out > 0 && cnt > 0 && H > 0, VP_ERR_BAD_ARG, "vp_gather_sum_rows: bad args");
  VP_REQUIRE(!(accumulate && !out_f32), VP_ERR_BAD_ARG, "vp_gather_sum_rows: accumulate needs an fp32 output");
  dim3 grid((H + 255) / 256, (unsigned)min(n_out, 8192L));
#define L(OF, SF) hipLaunchKernelGGL((gather_sum_rows_kernel<OF, SF>), grid, dim3(256), 0, s, src, lds, idx, cnt, scale, out, ldo, n_out, H, accumulate)
  if (out_f32 && src_f32) L(true, true);
  else if (out_f32) L(true, false);
  else if (src_f32) L(false, true);
  else L(false, false);
#undef L
  return vp_check_launch("vp_gather_sum_rows");
}

int vp_cast_f32_to_bf16(long n, const float* x, void* y, hipStream_t s) {
  VP_REQUIRE(n > 0, VP_ERR_BAD_ARG, "vp_cast_f32_to_bf16: bad n");
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, GRID_FOR(n), dim3(256), 0, s, x, (bf16_t*)y, n);
  return vp_check_launch("vp_cast_f32_to_bf16");
}

int vp_cast_bf16_to_f32(long n, const void* x, float* y, int accumulate, hipStream_t s) {
  VP_REQUIRE(n > 0, VP_ERR_BAD_ARG, "vp_cast_bf16_to_f32: bad n");
  hipLaunchKernelGGL(cast_bf16_to_f32_kernel, GRID_FOR(n), dim3(256), 0, s, (const bf16_t*)x, y, n, accumulate);
  return vp_check_launch("vp_cast_bf16_to_f32");
}

int vp_scatter_rows_bf16_to_f32(long n, int H, const void* src, long lds, const int* idx, float* dst, long ldd, hipStream_t s) {
  VP_REQUIRE(n > 0 && H > 0 && H % 8 == 0 && lds % 8 == 0 && ldd % 4 == 0 && src && dst, VP_ERR_BAD_ARG, "vp_scatter_rows_bf16_to_f32: bad args");
  hipLaunchKernelGGL(scatter_rows_bf16_to_f32_kernel, dim3((unsigned)min(8192L, (n + 3) / 4)), dim3(256), 0, s, (const bf16_t*)src, lds, idx, dst,
                     ldd, n, H);
  return vp_check_launch("vp_scatter_rows_bf16_to_f32");
}

int vp_sum_f32(long n, const float* x, float* out, float scale, hipStream_t s) {
  VP_REQUIRE(n > 0, VP_ERR_BAD_ARG, "vp_sum_f32: bad n");
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, s, x, out, n, scale);
  return vp_check_launch("vp_sum_f32");
}

int vp_scatter_add_rows(long n, int H, const void* src, long lds, const int* idx, float* dst, hipStream_t s) {
  VP_REQUIRE(n > 0 && H > 0 && H % 8 == 0 && lds % 8 == 0 && src && idx && dst, VP_ERR_BAD_ARG, "vp_scatter_add_rows: bad args");
  hipLaunchKernelGGL(scatter_add_rows_kernel, GRID_FOR(n * (H / 8)), dim3(256), 0, s, (const bf16_t*)src, idx, dst, n, H, lds);
  return vp_check_launch("vp_scatter_add_rows");
}

// out[0] = sum x^2; `part` = caller-owned workspace of vp_sumsq_nblk(n) floats
int vp_sumsq_nblk(long n) { return (int)max(1L, min(1024L, (n + 4095) / 4096)); }
int vp_sumsq_f32(long n, const float* x, float* part, float* out, hipStream_t s) {
  VP_REQUIRE(n > 0 && x && part && out, VP_ERR_BAD_ARG, "vp_sumsq_f32: bad args");
  const int nb = vp_sumsq_nblk(n);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, x, part, n);
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, s, part, out, (long)nb, 1.f);
  return vp_check_launch("vp_sumsq_f32");
}

int vp_adamw(long n, float* p, const float* g, float* m, float* v, void* bf16_shadow, float lr, float beta1, float beta2, float eps,
             float weight_decay, int step, float grad_scale, hipStream_t s) {
  VP_REQUIRE(n > 0 && step >= 1, VP_ERR_BAD_ARG, "vp_adamw: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  const long n4 = n >> 2;
  const bool vec = n4 > 0 && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0 && (((uintptr_t)bf16_shadow) & 7) == 0;
  if (vec) {
    hipLaunchKernelGGL(adamw4_kernel, dim3((unsigned)min((long)4096, (n4 + 255) / 256)), dim3(256), 0, s, (f32x4*)p, (const f32x4*)g, (f32x4*)m, (f32x4*)v,
                       (bf16x4*)bf16_shadow, n4, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    const long done = n4 << 2;
    if (done < n)                                       // the last n & 3 elements
      hipLaunchKernelGGL(adamw_kernel, dim3(1), dim3(64), 0, s, p + done, g + done, m + done, v + done,
                         bf16_shadow ? (bf16_t*)bf16_shadow + done : nullptr, n - done, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    return vp_check_launch("vp_adamw");
  }
  hipLaunchKernelGGL(adamw_kernel, GRID_FOR(n), dim3(256), 0, s, p, g, m, v, (bf16_t*)bf16_shadow, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2s, grad_scale);
  return vp_check_launch("vp_adamw");
}

}  // extern "C"
