// Embedding-distillation loss of the PT step: _emb_loss (base_ola_vlm.py:289-320) + calculate_contrastive_loss
// (ola_utils.py:108-125) with dist_collect's rank-ordered target gather (ola_utils.py:96-106) as an input.
//
//   sl1  = mean_all( smooth_l1_{beta=1}(pred, tgt_local) * mask_b )
//   con  = w * mean_b( CE_b( min(exp(s),100) * p^_b . t^_j , label = rank*B + b ) ) * mean_b(mask_b)      (p^, t^ L2-normalised)
//   loss = sl1 + con
//
// HBM-bound ("the distillation-loss reduction" of the north star).  FORWARD = ONE launch:
//   * streaming pass: every wave owns 32-element k-steps of the feature axis; the B x Bw dot products p_b.t_j are
//     MFMA 16x16x32 bf16 tiles fed STRAIGHT from global memory (an MFMA A/B fragment is "row = lane & 15, 8 consecutive k at
//     (lane >> 4) * 8", i.e. a 16-byte load from a row-major [rows, D] matrix: no LDS staging, no transposes), fp32 accumulate;
//     |p|^2, |t|^2 by v_dot2_f32_bf16 on the same registers, the smooth-L1 sum from an (L1-resident) second read of the local target
//     row.  pred and the gathered targets are each read once: algorithmic bytes = 2*D*(B + Bw).
//   * deterministic two-level tree over the per-block partial statistics with the "last block done" pattern (device-scope ticket
//     counters; fixed summation order, so results are bitwise reproducible), then the B x Bw softmax / loss scalars / backward
//     coefficients in the last block.  No atomics on data, no second or third launch (round 1 needed three: 13 + 11.5 + 4 us).
// BACKWARD = one streaming launch: dpred[b,:] = g * ( a_b * clamp(p_b - t_own(b), -1, 1) + sum_j c_bj * t_j - e_b * p_b ).
// Any local batch B <= 64 and any gathered batch Bw <= 1024 (the reference's pretrain.sh runs 32 per device on 8 devices: Bw = 256).
#include "common.h"

namespace {

constexpr int G1 = 32;                 // blocks per first-level reduction group
constexpr int MAX_GROUPS = 1024;       // njc * ngrp
constexpr int N_SLOTS = 8;             // independent counter sets (one per stream hash) so that calls on different streams do not collide
__device__ unsigned g_counters[N_SLOTS][1 + MAX_GROUPS];   // zero at module load; every launch leaves its slot zeroed again

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;

struct ElArgs {
  const bf16_t* pred;
  const bf16_t* tgt;
  const float* mask;
  const float* logit_scale;
  float* out3;
  float* coef;
  float* part;       // [njc][nblk][NS]
  float* part2;      // [njc][ngrp][NS]
  float* fin;        // PT[B][Bw] | TT[Bw] | PP[B] | SL[B]
  long D;
  int B, Bw, rank, nblk, njc, ngrp, slot;
  float w_con;
};

__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float c) {
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
#pragma unroll
  for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, ua[i]), __builtin_bit_cast(bf2_t, ub[i]), c, false);
  return c;
}

__device__ __forceinline__ float smooth_l1_8(bf16x8 p, bf16x8 t, float c) {
  const u32x4 up = __builtin_bit_cast(u32x4, p), ut = __builtin_bit_cast(u32x4, t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float d0 = fabsf(__builtin_bit_cast(float, up[i] << 16) - __builtin_bit_cast(float, ut[i] << 16));
    const float d1 = fabsf(__builtin_bit_cast(float, up[i] & 0xffff0000u) - __builtin_bit_cast(float, ut[i] & 0xffff0000u));
    c += d0 < 1.f ? 0.5f * d0 * d0 : d0 - 0.5f;
    c += d1 < 1.f ? 0.5f * d1 * d1 : d1 - 0.5f;
  }
  return c;
}

// The B x Bw softmax, the three loss scalars and the backward coefficients (single block, after the statistics are final):
//   coef[0..B)            a_b   : d(loss)/d(sl1 elementwise term) = mask_b / (B*D)
//   coef[B..2B)           e_b   : sum_j c_bj * (p_b.t_j) / |p_b|^2
//   coef[2B..2B+B*Bw)     c_bj  : dL/dZ_bj * scale / (|p_b| |t_j|)
//   coef[2B+B*Bw]         d loss / d logit_scale parameter
// mask semantics: sl1 = mean_all(sl1_elem * mask_b); con = w * mean_b(CE_b) * mean_b(mask_b)  (the outer-product broadcast of
// base_ola_vlm.py:312-316).
__device__ void el_finalize(const ElArgs& a, float* ce, float* dce) {
  const int t = threadIdx.x, B = a.B, Bw = a.Bw;
  const float* PT = a.fin;
  const float* TT = a.fin + (long)B * Bw;
  const float* PP = TT + Bw;
  const float* SL = PP + B;
  const bool has_con = a.logit_scale != nullptr;
  float scale = 0.f, dscale_dls = 0.f;
  if (has_con) {
    const float e = __expf(a.logit_scale[0]);
    scale = fminf(e, 100.f);
    dscale_dls = e < 100.f ? e : 0.f;
  }
  float msum = 0.f;
  for (int b = 0; b < B; ++b) msum += a.mask[b];
  const float mmean = msum / (float)B;
  if (t < B) {
    a.coef[t] = a.mask[t] / ((float)B * (float)a.D);
    if (has_con) {
      const float np = fmaxf(sqrtf(PP[t]), 1e-12f);
      const float* pt = PT + (long)t * Bw;
      float mx = -1e30f;
      for (int j = 0; j < Bw; ++j) mx = fmaxf(mx, scale * pt[j] / (np * fmaxf(sqrtf(TT[j]), 1e-12f)));
      float se = 0.f;
      for (int j = 0; j < Bw; ++j) se += __expf(scale * pt[j] / (np * fmaxf(sqrtf(TT[j]), 1e-12f)) - mx);
      const float lse = mx + __logf(se);
      const int own = a.rank * B + t;
      ce[t] = lse - scale * pt[own] / (np * fmaxf(sqrtf(TT[own]), 1e-12f));
      float e_b = 0.f, dls = 0.f;
      const float gz = a.w_con * mmean / (float)B;                 // d con / d Z_bj = w * mmean / B * (softmax_bj - onehot)
      for (int j = 0; j < Bw; ++j) {
        const float nt = fmaxf(sqrtf(TT[j]), 1e-12f);
        const float z = scale * pt[j] / (np * nt);
        float dz = __expf(z - lse);
        if (j == own) dz -= 1.f;
        dz *= gz;
        const float cbj = dz * scale / (np * nt);
        a.coef[2 * B + (long)t * Bw + j] = cbj;
        e_b += cbj * pt[j] / (np * np);
        dls += dz * (scale > 0.f ? z / scale : 0.f) * dscale_dls;
      }
      a.coef[B + t] = e_b;
      dce[t] = dls;
    } else {
      a.coef[B + t] = 0.f;
      for (int j = 0; j < Bw; ++j) a.coef[2 * B + (long)t * Bw + j] = 0.f;
    }
  }
  __syncthreads();
  if (t == 0) {
    float s1 = 0.f, cm = 0.f, dl = 0.f;
    for (int b = 0; b < B; ++b) {
      s1 += SL[b] * a.mask[b];
      if (has_con) { cm += ce[b]; dl += dce[b]; }
    }
    s1 /= ((float)B * (float)a.D);
    const float con = has_con ? a.w_con * (cm / (float)B) * mmean : 0.f;
    a.out3[0] = s1 + con;
    a.out3[1] = s1;
    a.out3[2] = con;
    a.coef[2 * B + (long)B * Bw] = dl;
  }
}

// grid (nblk, njc): block x streams k-steps {x*4 + wave + i * nblk*4}; chunk jc = gathered targets [jc*NG*16, (jc+1)*NG*16).
// NPB = 16-row blocks of local predictions, NG = 16-row groups of gathered targets per chunk.
template <int NPB, int NG>
__global__ __launch_bounds__(256) void emb_loss_fwd_kernel(const ElArgs a) {
  constexpr int PB = NPB * 16, TC = NG * 16, NS = PB * TC + TC + 2 * PB;
  __shared__ float red[NS];
  __shared__ float ce[64], dce[64];
  __shared__ unsigned ticket;
  const int jc = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
  const bool first = jc == 0;
  const long D = a.D;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[NPB][NG];
  float tt[NG], pp[NPB], sl[NPB];
  const bf16_t* prow[NPB];
  const bf16_t* orow[NPB];
  const bf16_t* trow[NG];
  bool pok[NPB], tok[NG];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int row = pb * 16 + r;
    pok[pb] = row < a.B;
    const int rc = min(row, a.B - 1);
    prow[pb] = a.pred + (long)rc * D;
    orow[pb] = a.tgt + (long)(a.rank * a.B + rc) * D;
    pp[pb] = 0.f;
    sl[pb] = 0.f;
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) acc[pb][ng] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int j = jc * TC + ng * 16 + r;
    tok[ng] = j < a.Bw;
    trow[ng] = a.tgt + (long)min(j, a.Bw - 1) * D;
    tt[ng] = 0.f;
  }
  const long nsteps = (D + 31) >> 5;
#pragma unroll 2
  for (long s = (long)bx * 4 + wv; s < nsteps; s += (long)gridDim.x * 4) {
    const long off0 = s * 32 + g * 8;
    const bool ok = off0 < D;                                  // D % 8 == 0: a lane's 8-vector is wholly inside or outside
    const long off = ok ? off0 : 0;
    bf16x8 pa[NPB], tb[NG], ow[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) pa[pb] = *(const bf16x8*)(prow[pb] + off);
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) tb[ng] = *(const bf16x8*)(trow[ng] + off);
    if (first) {
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) ow[pb] = *(const bf16x8*)(orow[pb] + off);
    }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) pa[pb] = (ok && pok[pb]) ? pa[pb] : zero8;
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) tb[ng] = (ok && tok[ng]) ? tb[ng] : zero8;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int ng = 0; ng < NG; ++ng) acc[pb][ng] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[pb], tb[ng], acc[pb][ng], 0, 0, 0);
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) tt[ng] = dot8(tb[ng], tb[ng], tt[ng]);
    if (first) {
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        pp[pb] = dot8(pa[pb], pa[pb], pp[pb]);
        sl[pb] = (ok && pok[pb]) ? smooth_l1_8(pa[pb], ow[pb], sl[pb]) : sl[pb];
      }
    }
  }
  // lane partials of the per-row sums: fold the 4 k-groups of the wave
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) { tt[ng] += __shfl_xor(tt[ng], 16, 64); tt[ng] += __shfl_xor(tt[ng], 32, 64); }
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    pp[pb] += __shfl_xor(pp[pb], 16, 64); pp[pb] += __shfl_xor(pp[pb], 32, 64);
    sl[pb] += __shfl_xor(sl[pb], 16, 64); sl[pb] += __shfl_xor(sl[pb], 32, 64);
  }
  // the 4 waves add into LDS one after the other (fixed order); MFMA C layout: col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = (pb * 16 + g * 4 + i) * TC + ng * 16 + r;
            red[idx] = (w == 0 ? 0.f : red[idx]) + acc[pb][ng][i];
          }
      if (g == 0) {
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) { const int idx = PB * TC + ng * 16 + r; red[idx] = (w == 0 ? 0.f : red[idx]) + tt[ng]; }
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
          const int i0 = PB * TC + TC + pb * 16 + r, i1 = i0 + PB;
          red[i0] = (w == 0 ? 0.f : red[i0]) + pp[pb];
          red[i1] = (w == 0 ? 0.f : red[i1]) + sl[pb];
        }
      }
    }
    __syncthreads();
  }
  float* mine = a.part + ((long)jc * a.nblk + bx) * NS;
  for (int i = tid; i < NS; i += 256) mine[i] = red[i];
  // ---- level 1: the last block of each group of G1 sums the group's partials (fixed order)
  unsigned* cnt = g_counters[a.slot];
  const int grp = bx / G1, gsz = min(G1, a.nblk - grp * G1);
  __threadfence();
  __syncthreads();
  if (tid == 0) ticket = atomicAdd(&cnt[1 + jc * a.ngrp + grp], 1u);
  __syncthreads();
  if (ticket != (unsigned)(gsz - 1)) return;
  __threadfence();
  {
    const float* src = a.part + ((long)jc * a.nblk + (long)grp * G1) * NS;
    float* dst = a.part2 + ((long)jc * a.ngrp + grp) * NS;
    for (int i = tid; i < NS; i += 256) {
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < gsz; ++q) s += src[(long)q * NS + i];
      dst[i] = s;
    }
  }
  if (tid == 0) cnt[1 + jc * a.ngrp + grp] = 0;                  // re-arm for the next launch
  // ---- level 2: the last group reducer sums the group partials of every chunk and finishes
  __threadfence();
  __syncthreads();
  if (tid == 0) ticket = atomicAdd(&cnt[0], 1u);
  __syncthreads();
  if (ticket != (unsigned)(a.njc * a.ngrp - 1)) return;
  __threadfence();
  if (tid == 0) cnt[0] = 0;
  {
    const int B = a.B, Bw = a.Bw;
    float* PT = a.fin;
    float* TT = a.fin + (long)B * Bw;
    float* PP = TT + Bw;
    float* SL = PP + B;
    for (int c = 0; c < a.njc; ++c) {
      const float* src = a.part2 + (long)c * a.ngrp * NS;
      for (int i = tid; i < NS; i += 256) {
        float s = 0.f;
        for (int q = 0; q < a.ngrp; ++q) s += src[(long)q * NS + i];
        if (i < PB * TC) {
          const int b = i / TC, j = c * TC + i % TC;
          if (b < B && j < Bw) PT[(long)b * Bw + j] = s;
        } else if (i < PB * TC + TC) {
          const int j = c * TC + (i - PB * TC);
          if (j < Bw) TT[j] = s;
        } else if (c == 0) {
          const int k = i - PB * TC - TC;
          if (k < PB) { if (k < B) PP[k] = s; }
          else if (k - PB < B) SL[k - PB] = s;
        }
      }
    }
  }
  __threadfence();
  __syncthreads();
  el_finalize(a, ce, dce);
}

// dpred[b,d] = gout * ( a_b * clamp(p - t_own, -1, 1) + sum_j c_bj * t_j[d] - e_b * p[b,d] ); grid (feature slabs, ceil(B/8)):
// a block row handles 8 local samples (8 x 8 fp32 accumulators per lane) and streams all Bw gathered targets once.
__global__ __launch_bounds__(256) void emb_loss_bwd_kernel(const bf16_t* __restrict__ pred, const bf16_t* __restrict__ tgt_all,
                                                           const float* __restrict__ coef, bf16_t* __restrict__ dpred, int B,
                                                           int Bw, long D, int rank, float gout) {
  extern __shared__ float cs[];                                  // [8][Bw] c_bj of this block row | a[8] | e[8]
  const int b0 = blockIdx.y * 8, nb = min(8, B - b0);
  for (int i = threadIdx.x; i < 8 * Bw; i += 256) {
    const int bb = i / Bw, j = i % Bw;
    cs[i] = bb < nb ? coef[2 * B + (long)(b0 + bb) * Bw + j] : 0.f;
  }
  if (threadIdx.x < 8) {
    cs[8 * Bw + threadIdx.x] = threadIdx.x < nb ? coef[b0 + threadIdx.x] : 0.f;
    cs[8 * Bw + 8 + threadIdx.x] = threadIdx.x < nb ? coef[B + b0 + threadIdx.x] : 0.f;
  }
  __syncthreads();
  const long nvec = D >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    float acc[8][8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
#pragma unroll 4
    for (int j = 0; j < Bw; ++j) {
      const bf16x8 tv = *(const bf16x8*)(tgt_all + (long)j * D + i * 8);
      float tf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tf[e] = bf2f((bf16_t)tv[e]);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const float cbj = cs[b * Bw + j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] += cbj * tf[e];
      }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < nb) {
        const bf16x8 pv = *(const bf16x8*)(pred + (long)(b0 + b) * D + i * 8);
        const bf16x8 tv = *(const bf16x8*)(tgt_all + (long)(rank * B + b0 + b) * D + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pf = bf2f((bf16_t)pv[e]);
          const float df = fminf(fmaxf(pf - bf2f((bf16_t)tv[e]), -1.f), 1.f);
          o[e] = (short)f2bf(gout * (cs[8 * Bw + b] * df + acc[b][e] - cs[8 * Bw + 8 + b] * pf));
        }
        *(bf16x8*)(dpred + (long)(b0 + b) * D + i * 8) = o;
      }
  }
}

struct ElPlan {
  int npb, ng, njc, nblk, ngrp, ns;
};

ElPlan el_plan(int B, int Bw, long D) {
  ElPlan p;
  p.npb = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
  const int groups = (Bw + 15) / 16;
  p.ng = groups <= 1 ? 1 : (groups <= 2 ? 2 : (groups <= 4 ? 4 : 8));
  p.njc = (groups + p.ng - 1) / p.ng;
  const long nsteps = (D + 31) / 32;
  // >= 2 k-steps per wave, at most 1024 streaming blocks per chunk (4 per CU) and MAX_GROUPS first-level groups in total
  long nblk = max(1L, min(1024L, (nsteps + 7) / 8));
  while (p.njc * ((nblk + G1 - 1) / G1) > MAX_GROUPS) nblk /= 2;
  p.nblk = (int)nblk;
  p.ngrp = (p.nblk + G1 - 1) / G1;
  p.ns = p.npb * 16 * p.ng * 16 + p.ng * 16 + 2 * p.npb * 16;
  return p;
}

template <int NPB>
void el_launch(int ng, dim3 grid, hipStream_t s, const ElArgs& a) {
  switch (ng) {
    case 1: hipLaunchKernelGGL((emb_loss_fwd_kernel<NPB, 1>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((emb_loss_fwd_kernel<NPB, 2>), grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((emb_loss_fwd_kernel<NPB, 4>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((emb_loss_fwd_kernel<NPB, 8>), grid, dim3(256), 0, s, a); break;
  }
}

}  // namespace

extern "C" {

// fp32 workspace of vp_emb_loss_fwd, in floats: per-block partials + group partials + final statistics (contents need not be
// initialised; nothing is kept between calls).
long vp_emb_loss_workspace(int B, int Bw, long D) {
  if (B <= 0 || Bw <= 0 || D <= 0) return 0;
  const ElPlan p = el_plan(B, Bw, D);
  return (long)p.njc * p.nblk * p.ns + (long)p.njc * p.ngrp * p.ns + (long)B * Bw + Bw + 2L * B + 64;
}

int vp_emb_loss_fwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* mask,
                    const float* logit_scale, float w_contrastive, float* out3, float* coef, float* workspace, hipStream_t s) {
  VP_REQUIRE(B > 0 && B <= 64 && Bw >= B && Bw <= 1024 && D > 0 && D % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_fwd: need 0<B<=64, B<=Bw<=1024, D%%8==0 (got B=%d Bw=%d D=%ld)", B, Bw, D);
  VP_REQUIRE(rank >= 0 && (long)(rank + 1) * B <= Bw, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: rank/B/Bw mismatch");
  VP_REQUIRE(pred && tgt_all && mask && out3 && coef && workspace, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: null pointer");
  VP_REQUIRE(((uintptr_t)pred | (uintptr_t)tgt_all) % 16 == 0, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: pred / tgt_all must be 16-byte aligned");
  const ElPlan p = el_plan(B, Bw, D);
  ElArgs a;
  a.pred = (const bf16_t*)pred; a.tgt = (const bf16_t*)tgt_all; a.mask = mask; a.logit_scale = logit_scale;
  a.out3 = out3; a.coef = coef;
  a.part = workspace;
  a.part2 = a.part + (long)p.njc * p.nblk * p.ns;
  a.fin = a.part2 + (long)p.njc * p.ngrp * p.ns;
  a.D = D; a.B = B; a.Bw = Bw; a.rank = rank; a.nblk = p.nblk; a.njc = p.njc; a.ngrp = p.ngrp;
  a.slot = (int)((((uintptr_t)s) >> 6) % N_SLOTS);
  a.w_con = w_contrastive;
  const dim3 grid(p.nblk, p.njc);
  if (p.npb == 1) el_launch<1>(p.ng, grid, s, a);
  else if (p.npb == 2) el_launch<2>(p.ng, grid, s, a);
  else el_launch<4>(p.ng, grid, s, a);
  return vp_check_launch("vp_emb_loss_fwd");
}

int vp_emb_loss_bwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* coef, float grad_out,
                    void* dpred, hipStream_t s) {
  VP_REQUIRE(B > 0 && B <= 64 && Bw >= B && Bw <= 1024 && D > 0 && D % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_bwd: need 0<B<=64, B<=Bw<=1024, D%%8==0 (got B=%d Bw=%d D=%ld)", B, Bw, D);
  VP_REQUIRE(rank >= 0 && (long)(rank + 1) * B <= Bw && pred && tgt_all && coef && dpred, VP_ERR_BAD_ARG, "vp_emb_loss_bwd: bad args");
  const int nby = (B + 7) / 8;
  const int nblk = (int)max(1L, min(2048L / nby, (D / 8 + 255) / 256));
  hipLaunchKernelGGL(emb_loss_bwd_kernel, dim3(nblk, nby), dim3(256), (8 * Bw + 16) * sizeof(float), s, (const bf16_t*)pred,
                     (const bf16_t*)tgt_all, coef, (bf16_t*)dpred, B, Bw, D, rank, grad_out);
  return vp_check_launch("vp_emb_loss_bwd");
}

}  // extern "C"
