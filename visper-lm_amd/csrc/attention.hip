// Flash-style softmax attention for the VisPer-LM hot path (forward + backward), bf16 MFMA 16x16x32.
//   * Llama/Phi-3 decoder: causal, GQA, D=128/96            (fwd + bwd; LLM is frozen -> dgrad only)
//   * CLIP-ViT: non-causal, N=577, D=64                      (fwd only)
//   * Perceiver resampler heads: cross-attention, D=32       (fwd + bwd)
// Design (CDNA4, 64-lane waves): every kernel keeps the *query* (or key) index on the lane axis so
// softmax statistics are lane-local, and builds the second GEMM's operand directly from the first GEMM's
// accumulator registers (no LDS round trip for P / dS): a 16x16x32 MFMA contracts over 32 "k slots"
// (g = lane>>4, j = 0..7); A and B only have to agree on which key each slot means, so slot (g,j) is
// mapped to key 16*(j>>2) + 4g + (j&3) — exactly the (row = 4g + r) layout two stacked C tiles have.
// The operand that is contracted over tokens (V in fwd, Q/dO in dK/dV, K in dQ) is staged *transposed*
// in LDS ([d][token], token pairs packed into dwords) so its fragments are two 8-byte reads.
// Row-major tiles are padded (+8 elements) -> conflict-free ds_read_b128; transposed tiles use a
// stride of 2*odd 8-byte slots -> conflict-free ds_read_b64.
// Log-sum-exp is kept in the log2 domain: lse2 = m + log2(l) with scores pre-multiplied by scale*log2(e).
#include "common.h"

struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  const bf16_t* dout; bf16_t* dq; bf16_t* dk; bf16_t* dv; float* delta;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  long do_bs, do_ts, dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
  int B, Hq, Hkv, Sq, Skv, window;
  const int* kv_len;
  float scale;
};

#define LOG2E 1.4426950408889634f
static __device__ __forceinline__ bf16x8 zero8() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// 8 fp32 (two C tiles' registers) -> one bf16x8 MFMA operand
static __device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
  r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
  return r;
}
// transposed-tile fragment: tokens (4g..4g+3) and (16+4g..) of feature row `d`
static __device__ __forceinline__ bf16x8 tfrag(const bf16_t* t, int ldt, int d, int tok0, int g) {
  const bf16x4 lo = *(const bf16x4*)(t + d * ldt + tok0 + 4 * g);
  const bf16x4 hi = *(const bf16x4*)(t + d * ldt + tok0 + 16 + 4 * g);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Stage R token rows x D features (global, token stride ts) into LDS row-major (stride D+8); rows >= limit -> 0.
template <int D, int R>
static __device__ __forceinline__ void stage_rows(bf16_t* lds, const bf16_t* gbase, long ts, int row0, int limit) {
  constexpr int CH = D / 8;
  for (int it = threadIdx.x; it < R * CH; it += 256) {
    const int r = it / CH, c = it % CH;
    bf16x8 v = zero8();
    if (row0 + r < limit) v = *(const bf16x8*)(gbase + (long)(row0 + r) * ts + c * 8);
    *(bf16x8*)(lds + r * (D + 8) + c * 8) = v;
  }
}
// Stage the same tile transposed: ldsT[d][token] (stride LDT), token pairs packed per dword.
// Optionally also writes the row-major image from the same loads (ROWMAJOR != nullptr).
template <int D, int R, int LDT>
static __device__ __forceinline__ void stage_transposed(bf16_t* ldsT, bf16_t* rowmajor, const bf16_t* gbase, long ts, int row0,
                                                        int limit) {
  constexpr int CH = D / 8, NP = R / 2;
  for (int it = threadIdx.x; it < NP * CH; it += 256) {
    const int pr = it % NP, c = it / NP;
    const int r0 = 2 * pr;
    bf16x8 a = zero8(), b = zero8();
    if (row0 + r0 < limit) a = *(const bf16x8*)(gbase + (long)(row0 + r0) * ts + c * 8);
    if (row0 + r0 + 1 < limit) b = *(const bf16x8*)(gbase + (long)(row0 + r0 + 1) * ts + c * 8);
    if (rowmajor) {
      *(bf16x8*)(rowmajor + r0 * (D + 8) + c * 8) = a;
      *(bf16x8*)(rowmajor + (r0 + 1) * (D + 8) + c * 8) = b;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t w = (uint32_t)(uint16_t)a[j] | ((uint32_t)(uint16_t)b[j] << 16);
      *(uint32_t*)(ldsT + (c * 8 + j) * LDT + r0) = w;
    }
  }
}

// ================================================================================================
// forward
// ================================================================================================
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  constexpr int LDK = D + 8, LDV = 72, NKS = D / 32, NDB = D / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * LDK];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[D * LDV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nqb = (p.Sq + 63) >> 6;
  const int qb = nqb - 1 - (int)blockIdx.x;            // heavy (late) causal blocks first
  const int h = blockIdx.y, b = blockIdx.z, hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 64, qrow = q0 + wave * 16 + fr;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;

  bf16x8 qf[NKS];
  {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_ts + (long)h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = (qrow < p.Sq) ? *(const bf16x8*)(qp + ks * 32 + g * 8) : zero8();
  }
  f32x4 oacc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -1e30f, l = 0.f;

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 64 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~63;
  const bf16_t* kbase = p.k + (long)b * p.k_bs + (long)hk * D;
  const bf16_t* vbase = p.v + (long)b * p.v_bs + (long)hk * D;

  for (int k0 = kstart; k0 < kend; k0 += 64) {
    __syncthreads();
    stage_rows<D, 64>(Ks, kbase, p.k_ts, k0, p.Skv);
    stage_transposed<D, 64, LDV>(Vt, nullptr, vbase, p.v_ts, k0, p.Skv);
    __syncthreads();

    f32x4 st[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 16 + fr) * LDK + ks * 32 + g * 8);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[kt], 0, 0, 0);
      }
    }
    // st[kt][r]: key = k0 + 16kt + 4g + r, query = qrow
    const bool need_mask = (k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > q0 + wave * 16 + off)) || (p.window > 0);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = st[kt][r] * c;
        if (need_mask) {
          const int key = k0 + kt * 16 + 4 * g + r;
          if (key >= kvlen || (CAUSAL && key > qrow + off) || (p.window > 0 && key <= qrow + off - p.window)) s = -INFINITY;
        }
        st[kt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(m, mx);
    const float alpha = exp2f(m - mnew);
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = exp2f(st[kt][r] - mnew);
        st[kt][r] = e;
        rs += e;
      }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mnew;
#pragma unroll
    for (int d = 0; d < NDB; ++d) oacc[d] *= alpha;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 pf = pack8(st[2 * kk], st[2 * kk + 1]);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const bf16x8 vf = tfrag(Vt, LDV, d * 16 + fr, kk * 32, g);
        oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[d], 0, 0, 0);
      }
    }
  }
  if (qrow < p.Sq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ts + (long)h * D;
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(oacc[d][r] * inv);
      *(bf16x4*)(op + d * 16 + 4 * g) = o;
    }
    if (p.lse && g == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = (l > 0.f) ? m + log2f(l) : -1e30f;
  }
}

// ================================================================================================
// backward pre-pass: delta[b,h,q] = sum_d dO * O
// ================================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
  const long rows = (long)p.B * p.Hq * p.Sq;
  const int lane = threadIdx.x & 63;
  for (long i = blockIdx.x * 4L + (threadIdx.x >> 6); i < rows; i += gridDim.x * 4L) {
    const int q = (int)(i % p.Sq);
    const int h = (int)((i / p.Sq) % p.Hq);
    const int b = (int)(i / ((long)p.Sq * p.Hq));
    const bf16_t* o = p.o + (long)b * p.o_bs + (long)q * p.o_ts + (long)h * D;
    const bf16_t* d = p.dout + (long)b * p.do_bs + (long)q * p.do_ts + (long)h * D;
    float a = 0.f;
    for (int e = lane * 2; e < D; e += 128) a += bf2f(o[e]) * bf2f(d[e]) + bf2f(o[e + 1]) * bf2f(d[e + 1]);
    a = wave_sum(a);
    if (lane == 0) p.delta[i] = a;
  }
}

// ================================================================================================
// backward: dK, dV  (block = 64 keys of one kv head; loops the GQA group's q heads and 32-query tiles)
// ================================================================================================
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnParams p) {
  constexpr int LDR = D + 8, LDT = 40, NKS = D / 32, NDB = D / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[32 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t dOs[32 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Qt[D * LDT];
  __shared__ __attribute__((aligned(16))) bf16_t dOt[D * LDT];
  __shared__ float lse_s[32], delta_s[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int hk = blockIdx.y, b = blockIdx.z;
  const int k0 = blockIdx.x * 64, key = k0 + wave * 16 + fr;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const int rep = p.Hq / p.Hkv;
  const float c = p.scale * LOG2E;

  bf16x8 kf[NKS], vf[NKS];
  {
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)key * p.k_ts + (long)hk * D;
    const bf16_t* vp = p.v + (long)b * p.v_bs + (long)key * p.v_ts + (long)hk * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[ks] = (key < p.Skv) ? *(const bf16x8*)(kp + ks * 32 + g * 8) : zero8();
      vf[ks] = (key < p.Skv) ? *(const bf16x8*)(vp + ks * 32 + g * 8) : zero8();
    }
  }
  f32x4 dk[NDB], dv[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  int qstart = 0, qend = p.Sq;
  if (CAUSAL) qstart = max(0, k0 - off) & ~31;
  if (p.window > 0) qend = min(p.Sq, k0 + 64 - off + p.window);
  if (k0 >= kvlen) qend = qstart;                      // whole key tile is padding: gradients are zero

  for (int hh = 0; hh < rep; ++hh) {
    const int h = hk * rep + hh;
    const bf16_t* qbase = p.q + (long)b * p.q_bs + (long)h * D;
    const bf16_t* dobase = p.dout + (long)b * p.do_bs + (long)h * D;
    const float* lse = p.lse + ((long)b * p.Hq + h) * p.Sq;
    const float* dl = p.delta + ((long)b * p.Hq + h) * p.Sq;
    for (int q0 = qstart; q0 < qend; q0 += 32) {
      __syncthreads();
      stage_transposed<D, 32, LDT>(Qt, Qs, qbase, p.q_ts, q0, p.Sq);
      stage_transposed<D, 32, LDT>(dOt, dOs, dobase, p.do_ts, q0, p.Sq);
      if (threadIdx.x < 32) {
        const int qq = q0 + threadIdx.x;
        lse_s[threadIdx.x] = qq < p.Sq ? lse[qq] : 0.f;
        delta_s[threadIdx.x] = qq < p.Sq ? dl[qq] : 0.f;
      }
      __syncthreads();
      f32x4 s[2], dp[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        s[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(Qs + (qt * 16 + fr) * LDR + ks * 32 + g * 8);
          const bf16x8 da = *(const bf16x8*)(dOs + (qt * 16 + fr) * LDR + ks * 32 + g * 8);
          s[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], s[qt], 0, 0, 0);
          dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dp[qt], 0, 0, 0);
        }
      }
      // s[qt][r]: query = q0 + 16qt + 4g + r, key = this lane's key
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qt * 16 + 4 * g + r, qg = q0 + ql;
          const bool ok = qg < p.Sq && key < kvlen && (!CAUSAL || key <= qg + off) &&
                          (p.window <= 0 || key > qg + off - p.window);
          const float pv = ok ? exp2f(s[qt][r] * c - lse_s[ql]) : 0.f;
          s[qt][r] = pv;
          dp[qt][r] = pv * (dp[qt][r] - delta_s[ql]);
        }
      const bf16x8 pf = pack8(s[0], s[1]);
      const bf16x8 dsf = pack8(dp[0], dp[1]);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const bf16x8 ta = tfrag(dOt, LDT, d * 16 + fr, 0, g);
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, pf, dv[d], 0, 0, 0);
        const bf16x8 tq = tfrag(Qt, LDT, d * 16 + fr, 0, g);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq, dsf, dk[d], 0, 0, 0);
      }
    }
  }
  if (key < p.Skv) {
    bf16_t* dkp = p.dk + (long)b * p.dk_bs + (long)key * p.dk_ts + (long)hk * D;
    bf16_t* dvp = p.dv + (long)b * p.dv_bs + (long)key * p.dv_ts + (long)hk * D;
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
      bf16x4 a, bb;
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = (short)f2bf(dk[d][r] * p.scale); bb[r] = (short)f2bf(dv[d][r]); }
      *(bf16x4*)(dkp + d * 16 + 4 * g) = a;
      *(bf16x4*)(dvp + d * 16 + 4 * g) = bb;
    }
  }
}

// ================================================================================================
// backward: dQ  (block = 64 queries of one q head; loops 64-key tiles)
// ================================================================================================
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr int LDR = D + 8, LDT = 72, NKS = D / 32, NDB = D / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Kt[D * LDT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nqb = (p.Sq + 63) >> 6;
  const int qb = nqb - 1 - (int)blockIdx.x;
  const int h = blockIdx.y, b = blockIdx.z, hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 64, qrow = q0 + wave * 16 + fr;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;

  bf16x8 qf[NKS], dof[NKS];
  {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_ts + (long)h * D;
    const bf16_t* dp_ = p.dout + (long)b * p.do_bs + (long)qrow * p.do_ts + (long)h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = (qrow < p.Sq) ? *(const bf16x8*)(qp + ks * 32 + g * 8) : zero8();
      dof[ks] = (qrow < p.Sq) ? *(const bf16x8*)(dp_ + ks * 32 + g * 8) : zero8();
    }
  }
  const long sidx = ((long)b * p.Hq + h) * p.Sq + qrow;
  const float lse = qrow < p.Sq ? p.lse[sidx] : 0.f;
  const float dlt = qrow < p.Sq ? p.delta[sidx] : 0.f;
  f32x4 dq[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 64 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~63;
  const bf16_t* kbase = p.k + (long)b * p.k_bs + (long)hk * D;
  const bf16_t* vbase = p.v + (long)b * p.v_bs + (long)hk * D;

  for (int k0 = kstart; k0 < kend; k0 += 64) {
    __syncthreads();
    stage_transposed<D, 64, LDT>(Kt, Ks, kbase, p.k_ts, k0, p.Skv);
    stage_rows<D, 64>(Vs, vbase, p.v_ts, k0, p.Skv);
    __syncthreads();
    f32x4 st[4], dpt[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dpt[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 ka = *(const bf16x8*)(Ks + (kt * 16 + fr) * LDR + ks * 32 + g * 8);
        const bf16x8 va = *(const bf16x8*)(Vs + (kt * 16 + fr) * LDR + ks * 32 + g * 8);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ks], st[kt], 0, 0, 0);
        dpt[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[ks], dpt[kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + kt * 16 + 4 * g + r;
        const bool ok = qrow < p.Sq && key < kvlen && (!CAUSAL || key <= qrow + off) &&
                        (p.window <= 0 || key > qrow + off - p.window);
        const float pv = ok ? exp2f(st[kt][r] * c - lse) : 0.f;
        dpt[kt][r] = pv * (dpt[kt][r] - dlt);
      }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 dsf = pack8(dpt[2 * kk], dpt[2 * kk + 1]);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const bf16x8 ktf = tfrag(Kt, LDT, d * 16 + fr, kk * 32, g);
        dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[d], 0, 0, 0);
      }
    }
  }
  if (qrow < p.Sq) {
    bf16_t* dqp = p.dq + (long)b * p.dq_bs + (long)qrow * p.dq_ts + (long)h * D;
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
      bf16x4 a;
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = (short)f2bf(dq[d][r] * p.scale);
      *(bf16x4*)(dqp + d * 16 + 4 * g) = a;
    }
  }
}

// ================================================================================================
// C ABI
// ================================================================================================
template <int D>
static int launch_fwd(const AttnParams& p, int causal, hipStream_t s) {
  dim3 grid((p.Sq + 63) / 64, p.Hq, p.B);
  if (causal) hipLaunchKernelGGL((attn_fwd_kernel<D, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<D, false>), grid, dim3(256), 0, s, p);
  return vp_check_launch("vp_attn_fwd");
}
template <int D>
static int launch_bwd(const AttnParams& p, int causal, hipStream_t s) {
  const long rows = (long)p.B * p.Hq * p.Sq;
  hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)min(8192L, (rows + 3) / 4)), dim3(256), 0, s, p);
  dim3 g1((p.Skv + 63) / 64, p.Hkv, p.B), g2((p.Sq + 63) / 64, p.Hq, p.B);
  if (causal) {
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, true>), g1, dim3(256), 0, s, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, true>), g2, dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, false>), g1, dim3(256), 0, s, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false>), g2, dim3(256), 0, s, p);
  }
  return vp_check_launch("vp_attn_bwd");
}

static int check_attn(const char* w, int B, int Hq, int Hkv, int Sq, int Skv, int D) {
  VP_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Sq > 0 && Skv > 0, VP_ERR_BAD_ARG, "%s: bad dims", w);
  VP_REQUIRE(Hq % Hkv == 0, VP_ERR_BAD_ARG, "%s: Hq %% Hkv != 0", w);
  VP_REQUIRE(D == 32 || D == 64 || D == 96 || D == 128, VP_ERR_UNSUPPORTED_SHAPE, "%s: head_dim %d not in {32,64,96,128}", w, D);
  return VP_OK;
}

extern "C" {

// Tensors are [B, S, H, D] views: element (b,s,h,d) at base + b*bs + s*ts + h*D + d (strides in elements,
// multiples of 8; bases 16-byte aligned).  lse: fp32 [B, Hq, Sq] (log2 domain).  kv_len: int32 [B] or NULL.
// causal: key j visible to query i iff j <= i + (Skv - Sq).  window > 0: additionally j > i + (Skv-Sq) - window.
int vp_attn_fwd(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k, long k_bs,
                long k_ts, const void* v, long v_bs, long v_ts, void* o, long o_bs, long o_ts, float* lse, const int* kv_len,
                int causal, int window, float scale, hipStream_t s) {
  int e = check_attn("vp_attn_fwd", B, Hq, Hkv, Sq, Skv, D);
  if (e) return e;
  AttnParams p{};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
  p.q_bs = q_bs; p.q_ts = q_ts; p.k_bs = k_bs; p.k_ts = k_ts; p.v_bs = v_bs; p.v_ts = v_ts; p.o_bs = o_bs; p.o_ts = o_ts;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv; p.window = window; p.kv_len = kv_len; p.scale = scale;
  switch (D) {
    case 32: return launch_fwd<32>(p, causal, s);
    case 64: return launch_fwd<64>(p, causal, s);
    case 96: return launch_fwd<96>(p, causal, s);
    default: return launch_fwd<128>(p, causal, s);
  }
}

// delta: fp32 workspace [B, Hq, Sq].  dq/dk/dv use the same [B,S,H,D] addressing with their own strides.
int vp_attn_bwd(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k, long k_bs,
                long k_ts, const void* v, long v_bs, long v_ts, const void* o, long o_bs, long o_ts, const float* lse,
                const void* dout, long do_bs, long do_ts, void* dq, long dq_bs, long dq_ts, void* dk, long dk_bs, long dk_ts,
                void* dv, long dv_bs, long dv_ts, float* delta, const int* kv_len, int causal, int window, float scale,
                hipStream_t s) {
  int e = check_attn("vp_attn_bwd", B, Hq, Hkv, Sq, Skv, D);
  if (e) return e;
  VP_REQUIRE(lse && delta && dout && dq && dk && dv, VP_ERR_BAD_ARG, "vp_attn_bwd: null pointer");
  AttnParams p{};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = (float*)lse;
  p.dout = (const bf16_t*)dout; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.delta = delta;
  p.q_bs = q_bs; p.q_ts = q_ts; p.k_bs = k_bs; p.k_ts = k_ts; p.v_bs = v_bs; p.v_ts = v_ts; p.o_bs = o_bs; p.o_ts = o_ts;
  p.do_bs = do_bs; p.do_ts = do_ts; p.dq_bs = dq_bs; p.dq_ts = dq_ts; p.dk_bs = dk_bs; p.dk_ts = dk_ts; p.dv_bs = dv_bs; p.dv_ts = dv_ts;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv; p.window = window; p.kv_len = kv_len; p.scale = scale;
  switch (D) {
    case 32: return launch_bwd<32>(p, causal, s);
    case 64: return launch_bwd<64>(p, causal, s);
    case 96: return launch_bwd<96>(p, causal, s);
    default: return launch_bwd<128>(p, causal, s);
  }
}

}  // extern "C"
