// Loss kernels of the PT train step.
//  (1) NTP cross-entropy over a chunk of bf16 logits (lm_head GEMM output): per-row fp32 online
//      softmax -> row loss, and dlogits written IN PLACE (bf16) for the dgrad GEMM.   [ola_llama.py:121-136]
//  (2) the embedding-distillation loss lives in emb_loss.hip.
#include "common.h"

// ---------------------------------------------------------------- NTP cross entropy ------------
// logits: [rows, V] bf16 (ld), labels int64 [rows] (-100 = ignore). row_loss[r] = lse - logit[label] (0 if ignored).
// dlogits (in place) = (softmax - onehot) * gscale  (0 for ignored rows).  One block per row.
__global__ __launch_bounds__(512) void ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, const long* __restrict__ labels,
                                                         float* __restrict__ row_loss, int V, long ld, float gscale,
                                                         int write_grad) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  bf16_t* lr = logits + row * ld;
  const long label = labels[row];
  const int nv = V >> 3;
  if (label < 0) {
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    if (write_grad) {
      for (int i = threadIdx.x; i < nv; i += 512) *(bf16x8*)(lr + i * 8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = nv * 8 + threadIdx.x; i < V; i += 512) lr[i] = 0;
    }
    return;
  }
  // pass 1: online max / sum-exp per thread, then block combine
  float m = -1e30f, s = 0.f;
  for (int i = threadIdx.x; i < nv; i += 512) {
    const bf16x8 v = *(const bf16x8*)(lr + i * 8);
    float mx = bf2f((bf16_t)v[0]);
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f((bf16_t)v[j]));
    if (mx > m) { s *= __expf(m - mx); m = mx; }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __expf(bf2f((bf16_t)v[j]) - m);
  }
  for (int i = nv * 8 + threadIdx.x; i < V; i += 512) {
    const float x = bf2f(lr[i]);
    if (x > m) { s *= __expf(m - x); m = x; }
    s += __expf(x - m);
  }
  const float M = block_max(m, red);
  const float S = block_sum(s * __expf(m - M), red);
  const float lse = M + __logf(S);
  if (threadIdx.x == 0) row_loss[row] = lse - bf2f(lr[label]);
  if (!write_grad) return;
  __syncthreads();           // label logit read before it is overwritten
  const float inv = gscale / S;
  for (int i = threadIdx.x; i < nv; i += 512) {
    const bf16x8 v = *(const bf16x8*)(lr + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gr = __expf(bf2f((bf16_t)v[j]) - M) * inv;
      if (i * 8 + j == label) gr -= gscale;
      o[j] = (short)f2bf(gr);
    }
    *(bf16x8*)(lr + i * 8) = o;
  }
  for (int i = nv * 8 + threadIdx.x; i < V; i += 512) {
    float gr = __expf(bf2f(lr[i]) - M) * inv;
    if (i == label) gr -= gscale;
    lr[i] = f2bf(gr);
  }
}

// The same kernel with the row held in REGISTERS between the two passes (round 5): NCH 16-byte chunks per thread (V <= NCH * 4096, V % 8 == 0), all loads
// in flight at once; the row is read from memory once instead of twice (V = 128256: 250 KB per row, 2.9 GB per step at configs[1]).  Same per-thread
// chunk order, same block reductions: bit-identical to ce_fwd_bwd_kernel.
template <int NCH>
__global__ __launch_bounds__(512) void ce_fwd_bwd_reg_kernel(bf16_t* __restrict__ logits, const long* __restrict__ labels,
                                                             float* __restrict__ row_loss, int V, long ld, float gscale, int write_grad) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  bf16_t* lr = logits + row * ld;
  const long label = labels[row];
  const int nv = V >> 3;
  if (label < 0) {
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    if (write_grad)
      for (int i = threadIdx.x; i < nv; i += 512) *(bf16x8*)(lr + i * 8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    return;
  }
  bf16x8 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = c * 512 + (int)threadIdx.x;
    if (i < nv) v[c] = *(const bf16x8*)(lr + i * 8);
  }
  float m = -1e30f, s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * 512 + (int)threadIdx.x < nv) {
      float mx = bf2f((bf16_t)v[c][0]);
#pragma unroll
      for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f((bf16_t)v[c][j]));
      if (mx > m) { s *= __expf(m - mx); m = mx; }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += __expf(bf2f((bf16_t)v[c][j]) - m);
    }
  }
  const float M = block_max(m, red);
  const float S = block_sum(s * __expf(m - M), red);
  const float lse = M + __logf(S);
  if (threadIdx.x == 0) row_loss[row] = lse - bf2f(lr[label]);
  if (!write_grad) return;
  __syncthreads();           // label logit read before it is overwritten
  const float inv = gscale / S;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = c * 512 + (int)threadIdx.x;
    if (i < nv) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float gr = __expf(bf2f((bf16_t)v[c][j]) - M) * inv;
        if (i * 8 + j == label) gr -= gscale;
        o[j] = (short)f2bf(gr);
      }
      *(bf16x8*)(lr + i * 8) = o;
    }
  }
}

extern "C" {

int vp_ce_fwd_bwd(long rows, int V, void* logits, long ld, const long* labels, float* row_loss, float grad_scale, int write_grad,
                  hipStream_t s) {
  VP_REQUIRE(rows > 0 && V > 0 && logits && labels && row_loss, VP_ERR_BAD_ARG, "vp_ce_fwd_bwd: bad args");
  VP_REQUIRE(ld % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE, "vp_ce_fwd_bwd: ld must be a multiple of 8");
  static const bool reg_path = [] { const char* e = getenv("VP_CE_REG"); return !e || atoi(e) != 0; }();
  const int nch = ((V >> 3) + 511) / 512;
  if (reg_path && V % 8 == 0 && nch <= 32) {            // the row fits the block's registers: one read of the logits instead of two
#define VP_CE_REG(N) hipLaunchKernelGGL(ce_fwd_bwd_reg_kernel<N>, dim3((unsigned)rows), dim3(512), 0, s, (bf16_t*)logits, labels, row_loss, V, ld, grad_scale, write_grad)
    if (nch <= 8) VP_CE_REG(8); else if (nch <= 16) VP_CE_REG(16); else VP_CE_REG(32);
#undef VP_CE_REG
    return vp_check_launch("vp_ce_fwd_bwd");
  }
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)rows), dim3(512), 0, s, (bf16_t*)logits, labels, row_loss, V, ld, grad_scale,
                     write_grad);
  return vp_check_launch("vp_ce_fwd_bwd");
}

}  // extern "C"
