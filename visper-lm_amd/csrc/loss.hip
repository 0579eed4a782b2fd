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
                                                             float* __restrict__ part, int B, int Bw, long D, int rank, int nslot) {
  // lane = (k-vector slot, gathered target j): every thread owns ONE target column j of this chunk and all 8 local
  // predictions, i.e. 8 + 3 accumulators (pt[.][j], tt_j, pp_j, and the smooth-L1 sum of the local pair whose target is j).
  // The 8 lanes of a slot read the same pred vectors (one coalesced broadcast request), so pred and targets are each
  // streamed exactly once per chunk.
  __shared__ float wred[4][EL_STATS];
  const int jc = blockIdx.y;
  const int nj = min(EL_B, Bw - jc * EL_B);
  const int j = threadIdx.x & 7, slot = threadIdx.x >> 3;          // 32 k-vector slots per block
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool has_t = j < nj;
  const int bl = jc * EL_B + j - rank * B;                          // local sample whose own target is column j (or out of range)
  const bool has_pair = has_t && bl >= 0 && bl < B;
  const bf16_t* trow = tgt_all + (long)(jc * EL_B + (has_t ? j : 0)) * D;
  float pt[EL_B], tt = 0.f, pp = 0.f, sl = 0.f;
#pragma unroll
  for (int b = 0; b < EL_B; ++b) pt[b] = 0.f;
  const long nvec = D >> 3;
  for (long i = blockIdx.x * 32L + slot; i < nvec; i += gridDim.x * 32L) {
    bf16x8 tv = {0, 0, 0, 0, 0, 0, 0, 0}, pv[EL_B];
    if (has_t) tv = *(const bf16x8*)(trow + i * 8);
#pragma unroll
    for (int b = 0; b < EL_B; ++b) {
      pv[b] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (b < B) pv[b] = *(const bf16x8*)(pred + (long)b * D + i * 8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float tf = bf2f((bf16_t)tv[e]);
      tt += tf * tf;
      float pj = 0.f, pl = 0.f;                                     // pred of sample j (for pp_j) and of the paired sample bl
#pragma unroll
      for (int b = 0; b < EL_B; ++b) {
        const float pf = bf2f((bf16_t)pv[b][e]);
        pt[b] += pf * tf;
        pj = (b == j) ? pf : pj;
        pl = (b == bl) ? pf : pl;
      }
      pp += pj * pj;
      const float d = fabsf(pl - tf);                               // smooth-L1, beta = 1
      sl += d < 1.f ? 0.5f * d * d : d - 0.5f;
    }
  }
  if (!has_pair) sl = 0.f;
  // reduce over the 8 slots of the wave that share j (lane bits 3..5), then over the 4 waves through LDS
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int b = 0; b < EL_B; ++b) pt[b] += __shfl_xor(pt[b], o, 64);
    tt += __shfl_xor(tt, o, 64);
    pp += __shfl_xor(pp, o, 64);
    sl += __shfl_xor(sl, o, 64);
  }
  if (threadIdx.x < EL_STATS) wred[0][threadIdx.x] = wred[1][threadIdx.x] = wred[2][threadIdx.x] = wred[3][threadIdx.x] = 0.f;
  __syncthreads();
  if (lane < 8) {
#pragma unroll
    for (int b = 0; b < EL_B; ++b) wred[wv][b * EL_B + j] = pt[b];
    if (j < B) wred[wv][EL_B * EL_B + j] = pp;
    wred[wv][EL_B * EL_B + EL_B + j] = tt;
    if (has_pair) wred[wv][EL_B * EL_B + 2 * EL_B + bl] = sl;
  }
  __syncthreads();
  if (threadIdx.x < EL_STATS)
    part[((long)jc * nslot + blockIdx.x) * EL_STATS + threadIdx.x] =
        (wred[0][threadIdx.x] + wred[1][threadIdx.x]) + (wred[2][threadIdx.x] + wred[3][threadIdx.x]);
}

// Second level: block jc sums the nblk per-block partial rows of its chunk (fixed order -> deterministic) into the chunk's LAST
// slot part[jc][nslot-1][.].  Rows are pulled into LDS with wide independent loads (a serial walk over L2 latencies made the old
// single-block finalize the slowest kernel of the three), then 88 x 2 threads each add one padded LDS column half.
__global__ __launch_bounds__(256) void emb_loss_reduce_kernel(float* __restrict__ part, int nblk, int nslot) {
  __shared__ float rows[128][EL_STATS + 1];
  __shared__ float halves[2][EL_STATS];
  const int jc = blockIdx.x, t = threadIdx.x;
  const float* src = part + (long)jc * nslot * EL_STATS;
  const int st = t % EL_STATS, hf = t / EL_STATS;      // threads 0..175 do the column sums
  float acc = 0.f;
  for (int base = 0; base < nblk; base += 128) {
    const int n = min(128, nblk - base);
    for (int q = t; q < n * EL_STATS; q += 256) rows[q / EL_STATS][q % EL_STATS] = src[(long)base * EL_STATS + q];
    __syncthreads();
    if (hf < 2) {
      float a0 = 0.f, a1 = 0.f;
      for (int k = hf; k < n; k += 4) { a0 += rows[k][st]; if (k + 2 < n) a1 += rows[k + 2][st]; }
      acc += a0 + a1;
    }
    __syncthreads();
  }
  if (hf < 2) halves[hf][st] = acc;
  __syncthreads();
  if (t < EL_STATS) part[((long)jc * nslot + nslot - 1) * EL_STATS + t] = halves[0][t] + halves[1][t];
}

// Single block. Produces out3 = {emb_loss, sl1_loss, contrastive_loss} (already masked/weighted as the
// reference does, NOT multiplied by the task weight) and the backward coefficients:
//   coef[0..B)            a_b   : d(loss)/d(sl1 elementwise term) = mask_b / (B*D)
//   coef[B..2B)           e_b   : sum_j c_bj * (p_b.t_j) / |p_b|^2
//   coef[2B..2B+B*Bw)     c_bj  : dL/dZ_bj * scale / (|p_b| |t_j|)
//   coef[2B+B*Bw]         dlogit_scale (d loss / d log-scale parameter)
// mask semantics: sl1 = mean_all(sl1_elem * mask_b); con = w * mean_b(CE_b) * mean_b(mask_b)  (outer-product quirk).
__global__ __launch_bounds__(256) void emb_loss_finalize_kernel(const float* __restrict__ part, int nblk, int njc, int B, int Bw,
                                                               long D, int rank, const float* __restrict__ mask,
                                                               const float* __restrict__ logit_scale, float w_con,
                                                               float* __restrict__ out3, float* __restrict__ coef) {
  __shared__ float S[8][EL_STATS];                     // per j-chunk statistics (reduced by emb_loss_reduce_kernel into the last slot)
  __shared__ float pt[EL_B][64], pp[EL_B], tt[64], sl[EL_B], Z[EL_B][64], ce[EL_B], dce[EL_B];
  const int t = threadIdx.x;
  for (int q = t; q < njc * EL_STATS; q += 256) S[q / EL_STATS][q % EL_STATS] = part[((long)(q / EL_STATS) * nblk + nblk - 1) * EL_STATS + q % EL_STATS];
  __syncthreads();
  for (int idx = t; idx < B * Bw; idx += 256) {
    const int b = idx / Bw, j = idx % Bw, jc = j / EL_B, jj = j % EL_B;
    pt[b][j] = S[jc][b * EL_B + jj];
  }
  for (int j = t; j < Bw; j += 256) {
    const int jc = j / EL_B, jj = j % EL_B;
    tt[j] = S[jc][EL_B * EL_B + EL_B + jj];
  }
  if (t < B) {
    float s1 = 0.f;
    for (int jc = 0; jc < njc; ++jc) s1 += S[jc][EL_B * EL_B + 2 * EL_B + t];
    pp[t] = S[0][EL_B * EL_B + t];
    sl[t] = s1;
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
// slots per j-chunk in the workspace: up to 256 streaming blocks (32 k-vectors per block pass, >= 4 passes each) + 1 slot for
// the chunk's reduced statistics
int vp_emb_loss_nblk(long D) { return 1 + (int)max(1L, min(256L, (D / 8 + 127) / 128)); }

int vp_emb_loss_fwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* mask,
                    const float* logit_scale, float w_contrastive, float* out3, float* coef, float* part, hipStream_t s) {
  VP_REQUIRE(B > 0 && B <= EL_B && Bw >= B && Bw <= 64 && D > 0 && D % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_fwd: need 0<B<=8, B<=Bw<=64, D%%8==0 (got B=%d Bw=%d D=%ld)", B, Bw, D);
  VP_REQUIRE(rank >= 0 && (rank + 1) * B <= Bw, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: rank/B/Bw mismatch");
  const int nblk = vp_emb_loss_nblk(D), njc = (Bw + EL_B - 1) / EL_B;
  hipLaunchKernelGGL(emb_loss_stats_kernel, dim3(nblk - 1, njc), dim3(256), 0, s, (const bf16_t*)pred, (const bf16_t*)tgt_all, part, B, Bw,
                     D, rank, nblk);
  hipLaunchKernelGGL(emb_loss_reduce_kernel, dim3(njc), dim3(256), 0, s, part, nblk - 1, nblk);
  hipLaunchKernelGGL(emb_loss_finalize_kernel, dim3(1), dim3(256), 0, s, part, nblk, njc, B, Bw, D, rank, mask, logit_scale,
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
