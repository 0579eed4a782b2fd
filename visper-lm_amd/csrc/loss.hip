// Loss kernels of the PT train step.
//  (1) NTP cross-entropy over a chunk of bf16 logits (lm_head GEMM output): per-row fp32 online
//      softmax -> row loss, and dlogits written IN PLACE (bf16) for the dgrad GEMM.   [ola_llama.py:121-136]
//  (2) Embedding-distillation loss (_emb_loss + calculate_contrastive_loss): ONE pass over pred and the
//      (all-gathered) targets producing {sum smooth-L1, |p|^2, |t|^2, p.t_j}; a tiny finalize kernel
//      turns the statistics into the three loss scalars + the backward coefficients; a second streaming
//      pass writes dpred.  HBM-bound: algorithmic bytes fwd = 2*D*(B + Bw)... see DESIGN.md.
//      [base_ola_vlm.py:289-320, ola_utils.py:108-125]
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

// ---------------------------------------------------------------- embedding loss ---------------
// Statistics layout per j-chunk (8 gathered targets): [pt 8x8 | pp 8 | tt 8 | sl1 8] = 88 floats.
constexpr int EL_B = 8;            // max local batch per launch tile
constexpr int EL_STATS = EL_B * EL_B + 3 * EL_B;

// grid (nblk, njc): block handles feature slab and gathered-target chunk jc (targets jc*8 .. jc*8+7).
// part: [njc][nblk][EL_STATS].
__global__ __launch_bounds__(256) void emb_loss_stats_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ tgt_all,
                                                             float* __restrict__ part, int B, int Bw, long D, int rank) {
  __shared__ float red[16];
  const int jc = blockIdx.y;
  const int nj = min(EL_B, Bw - jc * EL_B);
  float pt[EL_B][EL_B], pp[EL_B], tt[EL_B], sl[EL_B];
#pragma unroll
  for (int b = 0; b < EL_B; ++b) {
    pp[b] = tt[b] = sl[b] = 0.f;
#pragma unroll
    for (int j = 0; j < EL_B; ++j) pt[b][j] = 0.f;
  }
  const long nvec = D >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    bf16x8 pv[EL_B], tv[EL_B];
#pragma unroll
    for (int b = 0; b < EL_B; ++b)
      if (b < B) pv[b] = *(const bf16x8*)(pred + (long)b * D + i * 8);
#pragma unroll
    for (int j = 0; j < EL_B; ++j)
      if (j < nj) tv[j] = *(const bf16x8*)(tgt_all + (long)(jc * EL_B + j) * D + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float pf[EL_B], tf[EL_B];
#pragma unroll
      for (int b = 0; b < EL_B; ++b) pf[b] = b < B ? bf2f((bf16_t)pv[b][e]) : 0.f;
#pragma unroll
      for (int j = 0; j < EL_B; ++j) tf[j] = j < nj ? bf2f((bf16_t)tv[j][e]) : 0.f;
#pragma unroll
      for (int b = 0; b < EL_B; ++b) {
        pp[b] += pf[b] * pf[b];
#pragma unroll
        for (int j = 0; j < EL_B; ++j) pt[b][j] += pf[b] * tf[j];
      }
#pragma unroll
      for (int j = 0; j < EL_B; ++j) tt[j] += tf[j] * tf[j];
      // smooth-L1 (beta = 1) between local pairs: target of local sample b is gathered index rank*B + b
#pragma unroll
      for (int b = 0; b < EL_B; ++b) {
        const int jj = rank * B + b - jc * EL_B;
        if (b < B && jj >= 0 && jj < nj) {
          float tsel = 0.f;
#pragma unroll
          for (int j = 0; j < EL_B; ++j) tsel = (j == jj) ? tf[j] : tsel;
          const float d = fabsf(pf[b] - tsel);
          sl[b] += d < 1.f ? 0.5f * d * d : d - 0.5f;
        }
      }
    }
  }
  float* out = part + ((long)jc * gridDim.x + blockIdx.x) * EL_STATS;
#pragma unroll
  for (int b = 0; b < EL_B; ++b) {
#pragma unroll
    for (int j = 0; j < EL_B; ++j) {
      const float v = block_sum(pt[b][j], red);
      if (threadIdx.x == 0) out[b * EL_B + j] = v;
    }
    const float a = block_sum(pp[b], red), t2 = block_sum(tt[b], red), s1 = block_sum(sl[b], red);
    if (threadIdx.x == 0) {
      out[EL_B * EL_B + b] = a;
      out[EL_B * EL_B + EL_B + b] = t2;
      out[EL_B * EL_B + 2 * EL_B + b] = s1;
    }
  }
}

// Single block. Produces out3 = {emb_loss, sl1_loss, contrastive_loss} (already masked/weighted as the
// reference does, NOT multiplied by the task weight) and the backward coefficients:
//   coef[0..B)            a_b   : d(loss)/d(sl1 elementwise term) = mask_b / (B*D)
//   coef[B..2B)           e_b   : sum_j c_bj * (p_b.t_j) / |p_b|^2
//   coef[2B..2B+B*Bw)     c_bj  : dL/dZ_bj * scale / (|p_b| |t_j|)
//   coef[2B+B*Bw]         dlogit_scale (d loss / d log-scale parameter)
// mask semantics: sl1 = mean_all(sl1_elem * mask_b); con = w * mean_b(CE_b) * mean_b(mask_b)  (outer-product quirk).
__global__ __launch_bounds__(64) void emb_loss_finalize_kernel(const float* __restrict__ part, int nblk, int njc, int B, int Bw,
                                                               long D, int rank, const float* __restrict__ mask,
                                                               const float* __restrict__ logit_scale, float w_con,
                                                               float* __restrict__ out3, float* __restrict__ coef) {
  __shared__ float pt[EL_B][64], pp[EL_B], tt[64], sl[EL_B], Z[EL_B][64], ce[EL_B], dce[EL_B];
  const int t = threadIdx.x;
  // reduce partials (deterministic order)
  for (int idx = t; idx < B * Bw; idx += 64) {
    const int b = idx / Bw, j = idx % Bw, jc = j / EL_B, jj = j % EL_B;
    float a = 0.f;
    for (int k = 0; k < nblk; ++k) a += part[((long)jc * nblk + k) * EL_STATS + b * EL_B + jj];
    pt[b][j] = a;
  }
  for (int j = t; j < Bw; j += 64) {
    const int jc = j / EL_B, jj = j % EL_B;
    float a = 0.f;
    for (int k = 0; k < nblk; ++k) a += part[((long)jc * nblk + k) * EL_STATS + EL_B * EL_B + EL_B + jj];
    tt[j] = a;
  }
  if (t < B) {
    float a = 0.f, s = 0.f;
    for (int k = 0; k < nblk; ++k) a += part[((long)0 * nblk + k) * EL_STATS + EL_B * EL_B + t];
    for (int jc = 0; jc < njc; ++jc)
      for (int k = 0; k < nblk; ++k) s += part[((long)jc * nblk + k) * EL_STATS + EL_B * EL_B + 2 * EL_B + t];
    pp[t] = a;
    sl[t] = s;
  }
  __syncthreads();
  float scale = 0.f, dscale_dls = 0.f;
  const bool has_con = logit_scale != nullptr;
  if (has_con) {
    const float e = __expf(logit_scale[0]);
    scale = fminf(e, 100.f);
    dscale_dls = e < 100.f ? e : 0.f;
  }
  float msum = 0.f;
  for (int b = 0; b < B; ++b) msum += mask[b];
  const float mmean = msum / (float)B;
  if (t < B && has_con) {
    const float np = fmaxf(sqrtf(pp[t]), 1e-12f);
    float mx = -1e30f;
    for (int j = 0; j < Bw; ++j) {
      const float z = scale * pt[t][j] / (np * fmaxf(sqrtf(tt[j]), 1e-12f));
      Z[t][j] = z;
      mx = fmaxf(mx, z);
    }
    float se = 0.f;
    for (int j = 0; j < Bw; ++j) se += __expf(Z[t][j] - mx);
    const float lse = mx + __logf(se);
    ce[t] = lse - Z[t][rank * B + t];
    // d con / d Z_bj = w * mmean / B * (softmax_bj - onehot)
    float e_b = 0.f, dls = 0.f;
    const float gz = w_con * mmean / (float)B;
    for (int j = 0; j < Bw; ++j) {
      float dz = __expf(Z[t][j] - lse);
      if (j == rank * B + t) dz -= 1.f;
      dz *= gz;
      const float nt = fmaxf(sqrtf(tt[j]), 1e-12f);
      const float cbj = dz * scale / (np * nt);
      coef[2 * B + t * Bw + j] = cbj;
      e_b += cbj * pt[t][j] / (np * np);
      dls += dz * (scale > 0.f ? Z[t][j] / scale : 0.f) * dscale_dls;
    }
    coef[B + t] = e_b;
    dce[t] = dls;
  }
  if (t < B) {
    coef[t] = mask[t] / ((float)B * (float)D);
    if (!has_con) {
      coef[B + t] = 0.f;
      for (int j = 0; j < Bw; ++j) coef[2 * B + t * Bw + j] = 0.f;
    }
  }
  __syncthreads();
  if (t == 0) {
    float s1 = 0.f, cm = 0.f, dl = 0.f;
    for (int b = 0; b < B; ++b) {
      s1 += sl[b] * mask[b];
      if (has_con) { cm += ce[b]; dl += dce[b]; }
    }
    s1 /= ((float)B * (float)D);
    const float con = has_con ? w_con * (cm / (float)B) * mmean : 0.f;
    out3[0] = s1 + con;
    out3[1] = s1;
    out3[2] = con;
    coef[2 * B + B * Bw] = dl;
  }
}

// dpred[b,d] = gout * ( a_b * clamp(p - t_local, -1, 1) + sum_j c_bj * t_j[d] - e_b * p[b,d] )
__global__ __launch_bounds__(256) void emb_loss_bwd_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ tgt_all,
                                                           const float* __restrict__ coef, bf16_t* __restrict__ dpred, int B,
                                                           int Bw, long D, int rank, float gout) {
  __shared__ float cs[EL_B * 64 + 2 * EL_B];
  for (int i = threadIdx.x; i < 2 * B + B * Bw; i += 256) cs[i] = coef[i];
  __syncthreads();
  const long nvec = D >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    float acc[EL_B][8];
#pragma unroll
    for (int b = 0; b < EL_B; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
    for (int j = 0; j < Bw; ++j) {
      const bf16x8 tv = *(const bf16x8*)(tgt_all + (long)j * D + i * 8);
      float tf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tf[e] = bf2f((bf16_t)tv[e]);
#pragma unroll
      for (int b = 0; b < EL_B; ++b)
        if (b < B) {
          const float cbj = cs[2 * B + b * Bw + j];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[b][e] += cbj * tf[e];
        }
    }
#pragma unroll
    for (int b = 0; b < EL_B; ++b)
      if (b < B) {
        const bf16x8 pv = *(const bf16x8*)(pred + (long)b * D + i * 8);
        const bf16x8 tv = *(const bf16x8*)(tgt_all + (long)(rank * B + b) * D + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pf = bf2f((bf16_t)pv[e]);
          const float df = fminf(fmaxf(pf - bf2f((bf16_t)tv[e]), -1.f), 1.f);
          o[e] = (short)f2bf(gout * (cs[b] * df + acc[b][e] - cs[B + b] * pf));
        }
        *(bf16x8*)(dpred + (long)b * D + i * 8) = o;
      }
  }
}

extern "C" {

int vp_ce_fwd_bwd(long rows, int V, void* logits, long ld, const long* labels, float* row_loss, float grad_scale, int write_grad,
                  hipStream_t s) {
  VP_REQUIRE(rows > 0 && V > 0 && logits && labels && row_loss, VP_ERR_BAD_ARG, "vp_ce_fwd_bwd: bad args");
  VP_REQUIRE(ld % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE, "vp_ce_fwd_bwd: ld must be a multiple of 8");
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)rows), dim3(512), 0, s, (bf16_t*)logits, labels, row_loss, V, ld, grad_scale,
                     write_grad);
  return vp_check_launch("vp_ce_fwd_bwd");
}

// workspace (fp32): part = njc * nblk * 88 floats with njc = ceil(Bw/8), nblk = vp_emb_loss_nblk(D);
// coef = 2B + B*Bw + 1 floats (kept for the backward).
int vp_emb_loss_nblk(long D) { return (int)max(1L, min(512L, (D / 8 + 255) / 256)); }

int vp_emb_loss_fwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* mask,
                    const float* logit_scale, float w_contrastive, float* out3, float* coef, float* part, hipStream_t s) {
  VP_REQUIRE(B > 0 && B <= EL_B && Bw >= B && Bw <= 64 && D > 0 && D % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_fwd: need 0<B<=8, B<=Bw<=64, D%%8==0 (got B=%d Bw=%d D=%ld)", B, Bw, D);
  VP_REQUIRE(rank >= 0 && (rank + 1) * B <= Bw, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: rank/B/Bw mismatch");
  const int nblk = vp_emb_loss_nblk(D), njc = (Bw + EL_B - 1) / EL_B;
  hipLaunchKernelGGL(emb_loss_stats_kernel, dim3(nblk, njc), dim3(256), 0, s, (const bf16_t*)pred, (const bf16_t*)tgt_all, part, B, Bw,
                     D, rank);
  hipLaunchKernelGGL(emb_loss_finalize_kernel, dim3(1), dim3(64), 0, s, part, nblk, njc, B, Bw, D, rank, mask, logit_scale,
                     w_contrastive, out3, coef);
  return vp_check_launch("vp_emb_loss_fwd");
}

int vp_emb_loss_bwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* coef, float grad_out,
                    void* dpred, hipStream_t s) {
  VP_REQUIRE(B > 0 && B <= EL_B && Bw >= B && Bw <= 64 && D > 0 && D % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE, "vp_emb_loss_bwd: bad shape");
  const int nblk = (int)max(1L, min(2048L, (D / 8 + 255) / 256));
  hipLaunchKernelGGL(emb_loss_bwd_kernel, dim3(nblk), dim3(256), 0, s, (const bf16_t*)pred, (const bf16_t*)tgt_all, coef,
                     (bf16_t*)dpred, B, Bw, D, rank, grad_out);
  return vp_check_launch("vp_emb_loss_bwd");
}

}  // extern "C"
