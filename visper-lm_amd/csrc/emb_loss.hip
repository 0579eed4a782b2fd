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
#include <type_traits>
#include <stdlib.h>

namespace {

#ifndef EL_G1
#define EL_G1 16
#endif
constexpr int G1 = EL_G1;              // blocks per first-level reduction group (and the most groups the second level sums per chunk)
constexpr int MAX_GROUPS = 1024;       // njc * ngrp
constexpr int MAX_TASKS = 8;           // distillation heads batched into one launch (vp_emb_loss_fwd_multi: blockIdx.z = task)
// Ticket counters of the "last block done" tree: MAX_TASKS x (1 + MAX_GROUPS) unsigned in a CALLER-owned block (`counters`,
// vp_emb_loss_counter_bytes()), zeroed once by the caller; every launch leaves it zeroed again.  The library owns no device memory.

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
  int B, Bw, rank, nblk, njc, ngrp, task;
  unsigned* cnt;     // this task's ticket counters: [1 + MAX_GROUPS], zero on entry, zero on exit
  float w_con;
  long long* dbg;    // dev aid: 8 wall-clock stamps (100 MHz) of the finishing block, or null
};

// device-coherent (agent-scope, sc1) accesses for data that crosses workgroups on different XCDs
__device__ __forceinline__ void cstore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float cload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte forms of the same device-coherent accesses (buffer ops with cache policy sc0 sc1 = aux 17): a quarter of the instructions and of
// the dependent round trips of the reduction tree.  The descriptor is built from a WAVE-UNIFORM, 16-byte aligned base; `elem` (multiple of 4
// floats, < 2^29) is the lane's element offset from it.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t crsrc(const float* uniform_base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 cload4(__amdgpu_buffer_rsrc_t rs, int elem) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, elem * 4, 0, 17));
}
__device__ __forceinline__ void cstore4(__amdgpu_buffer_rsrc_t rs, int elem, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, elem * 4, 0, 17);
}

// sum of the 8 products of two bf16x8 vectors, fp32 FMAs.  (v_dot2_f32_bf16 is NOT used: on gfx950 its sums of squares came out up
// to 13 % off in tools/emb_loss_debug.py, while the MFMA dot products from the same registers were exact to 1e-7.)
__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float c) {
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c = fmaf(__builtin_bit_cast(float, ua[i] << 16), __builtin_bit_cast(float, ub[i] << 16), c);
    c = fmaf(__builtin_bit_cast(float, ua[i] & 0xffff0000u), __builtin_bit_cast(float, ub[i] & 0xffff0000u), c);
  }
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
// Wave reductions for the finalize: the __shfl_xor butterflies of common.h compile to six dependent ds_bpermute round trips through the LDS
// crossbar (~0.35 us per reduction; the finalize chains 11 of them = most of its 4 us).  Here the 16-lane rows are folded with four DPP moves
// (xor 1, xor 2, half-row mirror, row mirror: register-to-register) and the four row totals are combined through v_readlane.
template <bool MAX>
__device__ __forceinline__ float el_wave_reduce(float v) {
  auto comb = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v = comb(v, dpp(v, std::integral_constant<int, 0xB1>{}));     // quad_perm [1,0,3,2]
  v = comb(v, dpp(v, std::integral_constant<int, 0x4E>{}));     // quad_perm [2,3,0,1]
  v = comb(v, dpp(v, std::integral_constant<int, 0x141>{}));    // row_half_mirror
  v = comb(v, dpp(v, std::integral_constant<int, 0x140>{}));    // row_mirror: every lane holds its 16-lane row total
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16)),
              r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return comb(comb(r0, r1), comb(r2, r3));
}
__device__ __forceinline__ float el_wave_sum(float v) { return el_wave_reduce<false>(v); }
__device__ __forceinline__ float el_wave_max(float v) { return el_wave_reduce<true>(v); }

__device__ void el_finalize(const ElArgs& a, float* PT, int ldpt, const float* TT, const float* PP, const float* SL, float* ce,
                            float* dce, float mask_pre, float ls_pre, bool pt_in_lds) {
  // Block-level sync points of this function.  With the statistics in LDS a barrier only has to cover LDS traffic: __syncthreads() would also
  // wait (vmcnt(0)) for the coefficient STORES issued just before it, a 1-2 us global round trip each time, on the critical path of the launch.
  auto sync = [&]() {
    if (pt_in_lds) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __syncthreads(); }
  };
  // 256 threads.  Phase A: every (b, j) logit in parallel, written IN PLACE of its dot product (PT is LDS when all gathered targets
  // fit one chunk, else the global workspace).  Phase B: one wave per local row, lane-strided over the gathered targets with shuffle
  // reductions.  (A serial loop per row cost ~170 ns per target: 30 us at Bw = 64.)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, B = a.B, Bw = a.Bw;
  const bool has_con = a.logit_scale != nullptr;
  float scale = 0.f, dscale_dls = 0.f;
  if (has_con) {                                                  // (mask / logit_scale were loaded at kernel entry: no memory round trip here)
    const float e = __expf(ls_pre);
    scale = fminf(e, 100.f);
    dscale_dls = e < 100.f ? e : 0.f;
  }
  // mask -> LDS once (ce / dce double as scratch until phase B): a serial loop of dependent global loads per thread cost 4 us
  if (t < 64) { ce[t] = mask_pre; dce[t] = 0.f; }
  sync();
  float msum = 0.f;
  for (int b = 0; b < B; ++b) msum += ce[b];
  const float mmean = msum / (float)B;
  const float mymask = t < 64 ? ce[t] : 0.f;
  sync();
  for (int idx = t; idx < B * Bw; idx += 256) {
    const int b = idx / Bw, j = idx - b * Bw;
    float* z = PT + (long)b * ldpt + j;
    *z = has_con ? scale * *z / (fmaxf(sqrtf(PP[b]), 1e-12f) * fmaxf(sqrtf(TT[j]), 1e-12f)) : 0.f;
  }
  if (t < B) {
    a.coef[t] = mymask / ((float)B * (float)a.D);
    if (!has_con) a.coef[B + t] = 0.f;
  }
  sync();
  for (int b = wv; b < B; b += 4) {
    const float* z = PT + (long)b * ldpt;
    float* cb = a.coef + 2 * B + (long)b * Bw;
    if (!has_con) {
      for (int j = lane; j < Bw; j += 64) cb[j] = 0.f;
      continue;
    }
    const float gz = a.w_con * mmean / (float)B;                   // d con / d Z_bj = w * mmean / B * (softmax_bj - onehot)
    const float np = fmaxf(sqrtf(PP[b]), 1e-12f);
    const int own = a.rank * B + b;
    float mx = -1e30f;
    for (int j = lane; j < Bw; j += 64) mx = fmaxf(mx, z[j]);
    mx = el_wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < Bw; j += 64) se += __expf(z[j] - mx);
    se = el_wave_sum(se);
    const float lse = mx + __logf(se);
    float e_b = 0.f, dls = 0.f;
    for (int j = lane; j < Bw; j += 64) {
      const float zj = z[j], nt = fmaxf(sqrtf(TT[j]), 1e-12f);
      float dz = __expf(zj - lse);
      if (j == own) dz -= 1.f;
      dz *= gz;
      const float cbj = dz * scale / (np * nt);
      cb[j] = cbj;
      e_b += cbj * (zj * nt / (scale * np));                        // = c_bj * (p_b . t_j) / |p_b|^2
      dls += dz * (zj / scale) * dscale_dls;
    }
    e_b = el_wave_sum(e_b);
    dls = el_wave_sum(dls);
    if (lane == 0) {
      ce[b] = lse - z[own];
      dce[b] = dls;
      a.coef[B + b] = e_b;
    }
  }
  sync();
  // masked smooth-L1 sum, CE sum, d/dlogit_scale sum: wave 0, fixed shuffle order
  if (wv == 0) {
    float s1 = lane < B ? SL[lane] * mymask : 0.f;
    float cm = (has_con && lane < B) ? ce[lane] : 0.f;
    float dl = (has_con && lane < B) ? dce[lane] : 0.f;
    s1 = el_wave_sum(s1); cm = el_wave_sum(cm); dl = el_wave_sum(dl);
    if (lane == 0) {
      s1 /= ((float)B * (float)a.D);
      const float con = has_con ? a.w_con * (cm / (float)B) * mmean : 0.f;
      a.out3[0] = s1 + con;
      a.out3[1] = s1;
      a.out3[2] = con;
      a.coef[2 * B + (long)B * Bw] = dl;
    }
  }
}

// grid (nblk, njc): block x streams k-steps {x*4 + wave + i * nblk*4}; chunk jc = gathered targets [jc*NG*16, (jc+1)*NG*16).
// NPB = 16-row blocks of local predictions, NG = 16-row groups of gathered targets per chunk.
struct ElMulti { ElArgs t[MAX_TASKS]; };

#ifndef EL_INFLIGHT
#define EL_INFLIGHT 16                 // 16-byte fragment loads a lane keeps in flight per round (dev knob: tools/emb_loss_sweep.sh)
#endif

// PACK (B <= 8 and Bw <= 8: the single-GPU step): a 16-row MFMA operand holds the 8 samples at TWO different places of the feature axis
// (rows 0-7: first half of the wave's k-step, rows 8-15: second half), so no lane loads a duplicate row; the two diagonal 8x8 blocks of the
// 16x16 product are the two halves' dot products (the off-diagonal blocks mix the halves and are dropped).
template <int NPB, int NG, bool PACK>
__device__ __forceinline__ void emb_loss_fwd_body(const ElMulti mt) {
  static_assert(!PACK || (NPB == 1 && NG == 1), "PACK is the 8 x 8 case");
  constexpr int PB = NPB * 16, TC = NG * 16, NS = PB * TC + TC + 2 * PB;
  // NV consecutive 16-byte vectors per lane and k-step: the four k-groups of a row then cover a whole 128-byte line (NV = 2) instead of
  // half of one (an MFMA only needs A and B to agree on the k order, so "lane g owns bytes [32g, 32g+32) of the line" is as good as any)
  constexpr int NV = (2 * NPB + NG) <= 8 ? 2 : 1;
  constexpr int KS = 32 * NV * (PACK ? 2 : 1);                   // feature elements per wave k-step
  constexpr bool PAR = NS * 16 <= 40 * 1024;                     // the 4 waves' partials side by side in LDS, summed in one pass
  // blockIdx.z = task (one launch for every distillation head of the step: same B / Bw, own D, pointers, workspace and ticket counters);
  // the grid is sized for the longest task, the other tasks' surplus blocks leave at once (they hold no tickets)
  // (static indices + selects: a DYNAMICALLY indexed array inside a by-value kernel argument returned wrong elements — the compiler bug round 2 met
  // in the GEMM's balancing arguments)
  ElArgs a = mt.t[0];
#pragma unroll
  for (int t = 1; t < MAX_TASKS; ++t)
    if ((int)blockIdx.z == t) a = mt.t[t];
  if ((int)blockIdx.x >= a.nblk || (int)blockIdx.y >= a.njc) return;
  __shared__ __attribute__((aligned(16))) float red[PAR ? 4 * NS : NS];
  __shared__ float ce[64], dce[64];
  __shared__ unsigned ticket;
  const int jc = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
  const int rs = PACK ? (r & 7) : r;                             // sample row of this lane inside a 16-row operand
  const int lo = (PACK ? (r >> 3) * 32 * NV : 0) + g * 8 * NV;   // lane's element offset inside the wave's k-step
  const bool first = jc == 0;
  const long long t_start = a.dbg ? wall_clock64() : 0;
  if (a.dbg && bx == 0 && jc == 0 && tid == 0) a.dbg[0] = t_start;
  // the finishing block's two scalar inputs, fetched by every block under the stream (two loads) instead of as a dependent round trip at the end
  float mask_pre = (tid < 64 && tid < a.B) ? a.mask[tid] : 0.f;
  float ls_pre = a.logit_scale ? a.logit_scale[0] : 0.f;
  asm volatile("" : "+v"(mask_pre), "+v"(ls_pre));
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
    const int row = pb * 16 + rs;
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
    const int j = jc * TC + ng * 16 + rs;
    tok[ng] = j < a.Bw;
    trow[ng] = a.tgt + (long)min(j, a.Bw - 1) * D;
    tt[ng] = 0.f;
  }
  const long nsteps = (D + KS - 1) / KS;
  const long stride = (long)a.nblk * 4;            // (not gridDim.x: the grid is sized for the longest task of the launch)
  // The local sample of pred row r is gathered target rank*B + r.  When rank*B is a multiple of 16 that row sits in THIS lane's
  // fragment of target group own_g + pb (chunk 0), so the smooth-L1 term needs no extra load; otherwise it is re-read (L1-resident).
  const int own_g = (a.rank * a.B) >> 4;
  const bool own_in_regs = first && ((a.rank * a.B) & 15) == 0 && own_g + NPB <= NG;
  // DEPTH k-steps per round: all their 16-byte fragment loads are issued before the first MFMA (registers: DEPTH * NV * (2 NPB + NG) * 4)
  constexpr int DEPTH = EL_INFLIGHT / ((2 * NPB + NG) * NV) > 0 ? EL_INFLIGHT / ((2 * NPB + NG) * NV) : 1;
  for (long s0 = (long)bx * 4 + wv; s0 < nsteps; s0 += DEPTH * stride) {
    bf16x8 pa[DEPTH][NV][NPB], tb[DEPTH][NV][NG], ow[DEPTH][NV][NPB];
    bool ok[DEPTH][NV];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const long off0 = (s0 + d * stride) * KS + lo + v * 8;
        ok[d][v] = off0 < D;                                     // D % 8 == 0: a lane's 8-vector is wholly inside or outside
        const long off = ok[d][v] ? off0 : 0;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) pa[d][v][pb] = *(const bf16x8*)(prow[pb] + off);
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) tb[d][v][ng] = *(const bf16x8*)(trow[ng] + off);
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) ow[d][v][pb] = (first && !own_in_regs) ? *(const bf16x8*)(orow[pb] + off) : zero8;
      }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) pa[d][v][pb] = (ok[d][v] && pok[pb]) ? pa[d][v][pb] : zero8;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) tb[d][v][ng] = (ok[d][v] && tok[ng]) ? tb[d][v][ng] : zero8;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
          for (int ng = 0; ng < NG; ++ng)
            acc[pb][ng] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[d][v][pb], tb[d][v][ng], acc[pb][ng], 0, 0, 0);
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) tt[ng] = dot8(tb[d][v][ng], tb[d][v][ng], tt[ng]);
        if (first) {
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            pp[pb] = dot8(pa[d][v][pb], pa[d][v][pb], pp[pb]);
            bf16x8 o = ow[d][v][pb];
            if (own_in_regs) {
#pragma unroll
              for (int ng = 0; ng < NG; ++ng) o = (ng == own_g + pb) ? tb[d][v][ng] : o;
            }
            sl[pb] = (ok[d][v] && pok[pb]) ? smooth_l1_8(pa[d][v][pb], o, sl[pb]) : sl[pb];
          }
        }
      }
  }
  const long long t_stream = a.dbg ? wall_clock64() : 0;
  // lane partials of the per-row sums: fold the 4 k-groups of the wave (PACK: and the two half-steps, rows r and r ^ 8)
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    tt[ng] += __shfl_xor(tt[ng], 16, 64); tt[ng] += __shfl_xor(tt[ng], 32, 64);
    if (PACK) tt[ng] += __shfl_xor(tt[ng], 8, 64);
  }
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    pp[pb] += __shfl_xor(pp[pb], 16, 64); pp[pb] += __shfl_xor(pp[pb], 32, 64);
    sl[pb] += __shfl_xor(sl[pb], 16, 64); sl[pb] += __shfl_xor(sl[pb], 32, 64);
    if (PACK) { pp[pb] += __shfl_xor(pp[pb], 8, 64); sl[pb] += __shfl_xor(sl[pb], 8, 64); }
  }
  if (PACK) {
    // MFMA C layout: col = lane & 15, row = (lane >> 4) * 4 + i.  Element (b, j) of the first half sits in lane (g = b >> 2, r = j), its
    // second-half partner (b + 8, j + 8) in lane (g + 2, r + 8) = lane + 40; everything else is cross-half and is zeroed.
    const bool keep = g < 2 && r < 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float other = __shfl(acc[0][0][i], (lane + 40) & 63, 64);
      acc[0][0][i] = keep ? acc[0][0][i] + other : 0.f;
    }
  }
  // cross-block layout of a partial: PT[PB][TC] | TT[TC] | PP[PB] | SL[PB]
  const __amdgpu_buffer_rsrc_t prs = crsrc(a.part + ((long)jc * a.nblk + bx) * NS);
  if (PAR) {
    // every wave drops its partial into its own LDS copy, then one pass adds the four in wave order and sends the block's partial on its way
    float* mine = red + wv * NS;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int ng = 0; ng < NG; ++ng)
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[(pb * 16 + g * 4 + i) * TC + ng * 16 + r] = acc[pb][ng][i];
    if (g == 0) {
#pragma unroll
      for (int ng = 0; ng < NG; ++ng) mine[PB * TC + ng * 16 + r] = tt[ng];
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        mine[PB * TC + TC + pb * 16 + r] = pp[pb];
        mine[PB * TC + TC + PB + pb * 16 + r] = sl[pb];
      }
    }
    __syncthreads();
    for (int i = tid; i < NS / 4; i += 256) {
      const f32x4* q = (const f32x4*)red + i;
      cstore4(prs, 4 * i, ((q[0] + q[NS / 4]) + q[2 * (NS / 4)]) + q[3 * (NS / 4)]);
    }
  } else {
    // the 4 waves add into LDS one after the other (fixed order)
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
    for (int i = tid; i < NS / 4; i += 256) cstore4(prs, 4 * i, *(const f32x4*)(red + 4 * i));
  }
  // Cross-block traffic (partials, tickets) goes through DEVICE-COHERENT accesses (agent-scope relaxed atomics = sc1 loads / stores
  // that bypass the per-XCD L2's non-coherent lines), ordered by vmcnt(0) + the workgroup barrier.  A __threadfence() here would
  // write back and invalidate the whole 4 MB L2 of the XCD once per wave: measured 118 us instead of ~10 for the depth loss.
  // ---- level 1: the last block of each group of G1 sums the group's partials (fixed order)
  // Memory-model note (MI355X_MICROARCH.md, "Valid forms": {sc0 sc1 stores and loads on both sides} + a drained flag): the partials leave as
  // write-through `sc0 sc1` stores, every wave drains its own stores with an EXPLICIT s_waitcnt vmcnt(0) (inline asm: not left to what the
  // compiler happens to emit for __syncthreads), the workgroup barrier collects the waves, and only then one lane takes the agent-scope
  // ticket; the reducer reads the partials with `sc0 sc1` loads that bypass its L1 and the (non-coherent) L2 lines.
  unsigned* cnt = a.cnt;
  const int grp = bx / G1, gsz = min(G1, a.nblk - grp * G1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (tid == 0) ticket = __hip_atomic_fetch_add(&cnt[1 + jc * a.ngrp + grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != (unsigned)(gsz - 1)) return;
  const long long t_ticket1 = a.dbg ? wall_clock64() : 0;
  const int B = a.B, Bw = a.Bw;
  // final statistics: in LDS (`red`, the partial's own layout) when one chunk holds every gathered target, else in the workspace
  const bool in_lds = a.njc == 1;
  float* PT = in_lds ? red : a.fin;
  const int ldpt = in_lds ? TC : Bw;
  float* TT = in_lds ? red + PB * TC : a.fin + (long)B * Bw;
  float* PP = in_lds ? TT + TC : TT + Bw;
  float* SL = in_lds ? PP + PB : PP + B;
  auto scatter = [&](int c, int i, float v) {                   // statistic i of target chunk c -> its place in PT | TT | PP | SL
    if (in_lds) { red[i] = v; return; }
    if (i < PB * TC) {
      const int b = i / TC, j = c * TC + i % TC;
      if (b < B && j < Bw) PT[(long)b * Bw + j] = v;
    } else if (i < PB * TC + TC) {
      const int j = c * TC + (i - PB * TC);
      if (j < Bw) TT[j] = v;
    } else if (c == 0) {
      const int k = i - PB * TC - TC;
      if (k < PB) { if (k < B) PP[k] = v; }
      else if (k - PB < B) SL[k - PB] = v;
    }
  };
  const bool single = a.njc * a.ngrp == 1;                       // one group in total: its reducer is the finisher
  {
    const __amdgpu_buffer_rsrc_t src = crsrc(a.part + ((long)jc * a.nblk + (long)grp * G1) * NS);
    const __amdgpu_buffer_rsrc_t dst = crsrc(a.part2 + ((long)jc * a.ngrp + grp) * NS);
    for (int i = tid; i < NS / 4; i += 256) {
      f32x4 v[G1];
#pragma unroll
      for (int q = 0; q < G1; ++q) v[q] = q < gsz ? cload4(src, q * NS + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < G1; ++q) sum += v[q];                  // same order per element as ever: q = 0, 1, 2, ...
      if (single) {
#pragma unroll
        for (int e = 0; e < 4; ++e) scatter(0, 4 * i + e, sum[e]);
      } else cstore4(dst, 4 * i, sum);
    }
  }
  if (tid == 0) __hip_atomic_store(&cnt[1 + jc * a.ngrp + grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
  if (!single) {
    // ---- level 2: the last group reducer sums the group partials of every chunk and finishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's group-partial stores (sc0 sc1) have left before the ticket
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(&cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(a.njc * a.ngrp - 1)) return;
    if (tid == 0) __hip_atomic_store(&cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int c = 0; c < a.njc; ++c) {
      const __amdgpu_buffer_rsrc_t src = crsrc(a.part2 + (long)c * a.ngrp * NS);
      for (int i = tid; i < NS / 4; i += 256) {
        f32x4 v[G1];
#pragma unroll
        for (int q = 0; q < G1; ++q) v[q] = q < a.ngrp ? cload4(src, q * NS + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};       // ngrp <= G1 (el_plan: nblk <= G1 * G1)
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < G1; ++q) sum += v[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) scatter(c, 4 * i + e, sum[e]);
      }
    }
  }
  if (in_lds) {                                                 // the statistics were written by this very block: LDS-only barrier (a full
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // __syncthreads would wait for the counter re-arm stores: a global round trip)
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();
  }
  const long long t_reduced = a.dbg ? wall_clock64() : 0;
  el_finalize(a, PT, ldpt, TT, PP, SL, ce, dce, mask_pre, ls_pre, in_lds);
  if (a.dbg && tid == 0) {
    a.dbg[1] = t_start; a.dbg[2] = t_stream; a.dbg[3] = t_ticket1; a.dbg[4] = t_reduced; a.dbg[5] = wall_clock64();
  }
}

// Two entry points of the same body: the small shapes (every configuration the PT recipes run) are held to 3 waves per SIMD, so that the three
// heads of a step (3 x 256 blocks in one launch) are all resident at once; the big-batch shapes keep the registers they need.
template <int NPB, int NG, bool PACK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void emb_loss_fwd_kernel3(const ElMulti mt) {
  emb_loss_fwd_body<NPB, NG, PACK>(mt);
}
template <int NPB, int NG, bool PACK>
__global__ __launch_bounds__(256) void emb_loss_fwd_kernel2(const ElMulti mt) {
  emb_loss_fwd_body<NPB, NG, PACK>(mt);
}
template <int NPB, int NG, bool PACK>
void el_launch1(dim3 grid, hipStream_t s, const ElMulti& a) {
  if constexpr (NPB * NG <= 8) hipLaunchKernelGGL((emb_loss_fwd_kernel3<NPB, NG, PACK>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((emb_loss_fwd_kernel2<NPB, NG, PACK>), grid, dim3(256), 0, s, a);
}

// dpred[b,d] = gout * ( a_b * clamp(p - t_own, -1, 1) + sum_j c_bj * t_j[d] - e_b * p[b,d] ); grid (feature slabs, ceil(B/8)):
// a block row handles 8 local samples (8 x 8 fp32 accumulators per lane) and streams all Bw gathered targets once.
struct ElBwd { const bf16_t* pred; const bf16_t* tgt_all; const float* coef; bf16_t* dpred; long D; float gout; int nblk; };
struct ElBwdMulti { ElBwd t[MAX_TASKS]; };

__global__ __launch_bounds__(256) void emb_loss_bwd_kernel(const ElBwdMulti mt, int B, int Bw, int rank) {
  ElBwd tk = mt.t[0];                                            // blockIdx.z = task (static indices + selects: see the forward kernel)
#pragma unroll
  for (int t = 1; t < MAX_TASKS; ++t)
    if ((int)blockIdx.z == t) tk = mt.t[t];
  if ((int)blockIdx.x >= tk.nblk) return;
  const bf16_t* __restrict__ pred = tk.pred;
  const bf16_t* __restrict__ tgt_all = tk.tgt_all;
  const float* __restrict__ coef = tk.coef;
  bf16_t* __restrict__ dpred = tk.dpred;
  const long D = tk.D;
  const float gout = tk.gout;
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
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += tk.nblk * 256L) {
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
  bool pack;
};

ElPlan el_plan(int B, int Bw, long D) {
  ElPlan p;
  p.npb = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
  const int groups = (Bw + 15) / 16;
  p.ng = groups <= 1 ? 1 : (groups <= 2 ? 2 : (groups <= 4 ? 4 : 8));
  p.njc = (groups + p.ng - 1) / p.ng;
  p.pack = B <= 8 && Bw <= 8;                                    // (then npb = ng = 1)
  const int nv = (2 * p.npb + p.ng) <= 8 ? 2 : 1;                // = the kernel's NV / KS
  const long ks = 32L * nv * (p.pack ? 2 : 1);
  const long nsteps = (D + ks - 1) / ks;
  // >= spw k-steps per wave, at most `cap` streaming blocks per chunk and MAX_GROUPS first-level groups in total
  static const long cap = getenv("VP_EL_NBLK") ? atol(getenv("VP_EL_NBLK")) : 256L;       // dev knob (tools/emb_loss_bench.py)
  static const long spw = getenv("VP_EL_SPW") ? atol(getenv("VP_EL_SPW")) : 2L;             // target k-steps per wave
  long nblk = max(1L, min(min(cap, (long)G1 * G1), (nsteps + 4 * spw - 1) / (4 * spw)));
  while (p.njc * ((nblk + G1 - 1) / G1) > MAX_GROUPS) nblk /= 2;
  p.nblk = (int)nblk;
  p.ngrp = (p.nblk + G1 - 1) / G1;
  p.ns = p.npb * 16 * p.ng * 16 + p.ng * 16 + 2 * p.npb * 16;
  return p;
}

template <int NPB>
void el_launch(int ng, dim3 grid, hipStream_t s, const ElMulti& a) {
  switch (ng) {
    case 1: el_launch1<NPB, 1, false>(grid, s, a); break;
    case 2: el_launch1<NPB, 2, false>(grid, s, a); break;
    case 4: el_launch1<NPB, 4, false>(grid, s, a); break;
    default: el_launch1<NPB, 8, false>(grid, s, a); break;
  }
}

}  // namespace

#ifdef VP_DEBUG
long long* g_dbg = nullptr;
#endif

extern "C" {

// dev aid (tools/emb_loss_debug.py): device buffer of 8 int64 receiving wall-clock stamps of the next vp_emb_loss_fwd calls (NULL = off)
#ifdef VP_DEBUG
int vp_debug_emb_loss_stamps(long long* dev_buf) { g_dbg = dev_buf; return VP_OK; }
#endif

// bytes of the caller-owned ticket-counter block of vp_emb_loss_fwd / _fwd_multi (zeroed once; left zeroed by every launch)
long vp_emb_loss_counter_bytes(void) { return (long)MAX_TASKS * (1 + MAX_GROUPS) * sizeof(unsigned); }

// fp32 workspace of vp_emb_loss_fwd, in floats: per-block partials + group partials + final statistics (contents need not be
// initialised; nothing is kept between calls).
long vp_emb_loss_workspace(int B, int Bw, long D) {
  if (B <= 0 || Bw <= 0 || D <= 0) return 0;
  const ElPlan p = el_plan(B, Bw, D);
  return (long)p.njc * p.nblk * p.ns + (long)p.njc * p.ngrp * p.ns + (long)B * Bw + Bw + 2L * B + 64;
}

// One launch for `ntask` distillation heads that share (B, Bw, rank): task t has its own feature length D[t], pointers, workspace
// (vp_emb_loss_workspace(B, Bw, D[t]) floats) and contrastive weight.  The reference calls _emb_loss once per head and layer
// (base_ola_vlm.py:445-534); batching them turns 2 x ntask tiny launches per step into 2, and a single launch streams all heads' bytes.
int vp_emb_loss_fwd_multi(int ntask, int B, int Bw, const long* D, int rank, const void* const* pred, const void* const* tgt_all,
                          const float* const* mask, const float* const* logit_scale, const float* w_contrastive, float* const* out3,
                          float* const* coef, float* const* workspace, unsigned* counters, hipStream_t s) {
  VP_REQUIRE(ntask >= 1 && ntask <= MAX_TASKS && D && pred && tgt_all && mask && logit_scale && w_contrastive && out3 && coef && workspace && counters,
             VP_ERR_BAD_ARG, "vp_emb_loss_fwd_multi: 1 <= ntask <= %d and non-null arrays / counters", MAX_TASKS);
  VP_REQUIRE(B > 0 && B <= 64 && Bw >= B && Bw <= 1024, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_fwd: need 0<B<=64, B<=Bw<=1024 (got B=%d Bw=%d)", B, Bw);
  VP_REQUIRE(rank >= 0 && (long)(rank + 1) * B <= Bw, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: rank/B/Bw mismatch");
  ElMulti m;
  ElPlan p0 = el_plan(B, Bw, D[0] > 0 ? D[0] : 8);
  int gx = 0;
  for (int t = 0; t < ntask; ++t) {
    VP_REQUIRE(D[t] > 0 && D[t] % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE, "vp_emb_loss_fwd: need D%%8==0 (got D=%ld)", D[t]);
    VP_REQUIRE(pred[t] && tgt_all[t] && mask[t] && out3[t] && coef[t] && workspace[t], VP_ERR_BAD_ARG, "vp_emb_loss_fwd: null pointer");
    VP_REQUIRE(((uintptr_t)pred[t] | (uintptr_t)tgt_all[t] | (uintptr_t)workspace[t]) % 16 == 0, VP_ERR_BAD_ARG,
               "vp_emb_loss_fwd: pred / tgt_all / workspace must be 16-byte aligned");
    const ElPlan p = el_plan(B, Bw, D[t]);                       // npb / ng / njc depend on (B, Bw) only: the same for every task
    ElArgs& a = m.t[t];
    a.pred = (const bf16_t*)pred[t]; a.tgt = (const bf16_t*)tgt_all[t]; a.mask = mask[t]; a.logit_scale = logit_scale[t];
    a.out3 = out3[t]; a.coef = coef[t];
    a.part = workspace[t];
    a.part2 = a.part + (long)p.njc * p.nblk * p.ns;
    a.fin = a.part2 + (long)p.njc * p.ngrp * p.ns;
    a.D = D[t]; a.B = B; a.Bw = Bw; a.rank = rank; a.nblk = p.nblk; a.njc = p.njc; a.ngrp = p.ngrp;
    a.task = t; a.cnt = counters + (long)t * (1 + MAX_GROUPS);
    a.w_con = w_contrastive[t];
#ifdef VP_DEBUG
    a.dbg = t == 0 ? g_dbg : nullptr;
#else
    a.dbg = nullptr;
#endif
    gx = p.nblk > gx ? p.nblk : gx;
    p0 = p;
  }
  const dim3 grid(gx, p0.njc, ntask);
  if (p0.pack) el_launch1<1, 1, true>(grid, s, m);
  else if (p0.npb == 1) el_launch<1>(p0.ng, grid, s, m);
  else if (p0.npb == 2) el_launch<2>(p0.ng, grid, s, m);
  else el_launch<4>(p0.ng, grid, s, m);
  return vp_check_launch("vp_emb_loss_fwd");
}

int vp_emb_loss_fwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* mask,
                    const float* logit_scale, float w_contrastive, float* out3, float* coef, float* workspace, unsigned* counters,
                    hipStream_t s) {
  VP_REQUIRE(B > 0 && B <= 64 && Bw >= B && Bw <= 1024, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_fwd: need 0<B<=64, B<=Bw<=1024 (got B=%d Bw=%d)", B, Bw);
  VP_REQUIRE(pred && tgt_all && mask && out3 && coef && workspace, VP_ERR_BAD_ARG, "vp_emb_loss_fwd: null pointer");
  return vp_emb_loss_fwd_multi(1, B, Bw, &D, rank, &pred, &tgt_all, &mask, &logit_scale, &w_contrastive, &out3, &coef, &workspace, counters, s);
}

int vp_emb_loss_bwd_multi(int ntask, int B, int Bw, const long* D, int rank, const void* const* pred, const void* const* tgt_all,
                          const float* const* coef, const float* grad_out, void* const* dpred, hipStream_t s) {
  VP_REQUIRE(ntask >= 1 && ntask <= MAX_TASKS && D && pred && tgt_all && coef && grad_out && dpred, VP_ERR_BAD_ARG,
             "vp_emb_loss_bwd_multi: 1 <= ntask <= %d and non-null arrays", MAX_TASKS);
  VP_REQUIRE(B > 0 && B <= 64 && Bw >= B && Bw <= 1024, VP_ERR_UNSUPPORTED_SHAPE,
             "vp_emb_loss_bwd: need 0<B<=64, B<=Bw<=1024 (got B=%d Bw=%d)", B, Bw);
  VP_REQUIRE(rank >= 0 && (long)(rank + 1) * B <= Bw, VP_ERR_BAD_ARG, "vp_emb_loss_bwd: bad args");
  const int nby = (B + 7) / 8;
  ElBwdMulti m;
  int gx = 0;
  for (int t = 0; t < ntask; ++t) {
    VP_REQUIRE(D[t] > 0 && D[t] % 8 == 0, VP_ERR_UNSUPPORTED_SHAPE, "vp_emb_loss_bwd: need D%%8==0 (got D=%ld)", D[t]);
    VP_REQUIRE(pred[t] && tgt_all[t] && coef[t] && dpred[t], VP_ERR_BAD_ARG, "vp_emb_loss_bwd: bad args");
    const int nblk = (int)max(1L, min(2048L / nby, (D[t] / 8 + 255) / 256));
    m.t[t] = ElBwd{(const bf16_t*)pred[t], (const bf16_t*)tgt_all[t], coef[t], (bf16_t*)dpred[t], D[t], grad_out[t], nblk};
    gx = nblk > gx ? nblk : gx;
  }
  hipLaunchKernelGGL(emb_loss_bwd_kernel, dim3(gx, nby, ntask), dim3(256), (8 * Bw + 16) * sizeof(float), s, m, B, Bw, rank);
  return vp_check_launch("vp_emb_loss_bwd");
}

int vp_emb_loss_bwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* coef, float grad_out,
                    void* dpred, hipStream_t s) {
  VP_REQUIRE(pred && tgt_all && coef && dpred, VP_ERR_BAD_ARG, "vp_emb_loss_bwd: bad args");
  return vp_emb_loss_bwd_multi(1, B, Bw, &D, rank, &pred, &tgt_all, &coef, &grad_out, &dpred, s);
}

}  // extern "C"
