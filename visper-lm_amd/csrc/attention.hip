#include "attention_common.h"

// ================================================================================================
// forward: block = 128 queries (8 waves x 16), loops 64-key tiles
// ================================================================================================
// dev aid (tools/attn_phase_stamps.py, VP_ATTN_DBG=1): per-phase shader-cycle sums of waves 0 and 7 of the heaviest block, [wave][phase]
__device__ long vp_attn_dbg[2 * 8];
#define ATTN_STAMP(I) if (DBG && dbg_on) { const long t_ = clock64(); dbg_acc[I] += t_ - dbg_t; dbg_t = t_; }
template <int D, bool CAUSAL, bool BIAS = false, bool DBG = false>   // BIAS: separate instantiation (the extra live pointers cost the plain path 40 %)
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnParams p) {
  constexpr int LD = D + 16, NKS = D / 32, NDB = D / 16, TILE = 64 * LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  bf16_t* const Kbuf = (bf16_t*)attn_smem;             // [2][64*LD]   K tiles (double buffered)
  bf16_t* const Vbuf = Kbuf + 2 * TILE;                // [2][64*LD]   V tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nqb = (p.Sq + 127) >> 7;
  const int qb = nqb - 1 - VP_BZ(p);            // z is the slowest dispatch index: heavy (late) causal blocks first
  // blocks are dealt round-robin to the 8 XCDs (x & 7): give each XCD whole GQA groups so K/V tiles are shared in its L2
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 128, qw0 = q0 + wave * 16;
  const int qrow = qw0 + fr;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;

  bf16x8 qf[NKS];
  {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)min(qrow, p.Sq - 1) * p.q_ts + (long)h * D;   // clamped: rows >= Sq never stored
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32 + g * 8);
  }
  f32x4 oacc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -1e30f, l = 0.f;          // m is kept PRE-scaled: m = c * max(raw score)

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 128 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~63;
  const bf16_t* kbase = p.k + (long)b * p.k_bs + (long)hk * D;
  const bf16_t* vbase = p.v + (long)b * p.v_bs + (long)hk * D;

  TileRegs<D, 64> kr, vr;
  const __amdgpu_buffer_rsrc_t krs = TileRegs<D, 64>::rsrc(kbase, p.k_ts, p.Skv), vrs = TileRegs<D, 64>::rsrc(vbase, p.v_ts, p.Skv);
  int koff[TileRegs<D, 64>::N], vofs[TileRegs<D, 64>::N];
  kr.voff((int)p.k_ts, koff);
  vr.voff((int)p.v_ts, vofs);
  if (kstart < kend) {
    kr.load_buf(krs, koff, kstart, (int)p.k_ts);
    vr.load_buf(vrs, vofs, kstart, (int)p.v_ts);
    kr.store(Kbuf, LD);
    vr.store(Vbuf, LD);
  }
  __syncthreads();
  const bool dbg_on = DBG && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave == 0 || wave == 7) && lane == 0;
  long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = DBG ? clock64() : 0;
  int cur = 0;
  for (int k0 = kstart; k0 < kend; k0 += 64, cur ^= 1) {
    const bool more = k0 + 64 < kend;
    if (more) {
      kr.load_buf(krs, koff, k0 + 64, (int)p.k_ts);
      vr.load_buf(vrs, vofs, k0 + 64, (int)p.v_ts);
    }
    ATTN_STAMP(0)                                      // issue of the next tile's loads
    const bf16_t* Ks = Kbuf + cur * TILE;
    const bf16_t* Vs = Vbuf + cur * TILE;
    // wave-uniform skip of tiles that are entirely above this wave's causal diagonal
    const bool active = !CAUSAL || (k0 <= qw0 + 15 + off);
    if (active) {
      f32x4 st[4];
      __builtin_amdgcn_s_setprio(1);                  // MFMA bursts at raised priority (4 waves per SIMD share the pipe): -2.6 %
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 16 + fr) * LD + ks * 32 + g * 8);
          st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[kt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (DBG) { asm volatile("" :: "v"(st[0]), "v"(st[1]), "v"(st[2]), "v"(st[3])); asm volatile("s_nop 0" ::: "memory"); }
      ATTN_STAMP(1)                                    // S = K Q^T done (results consumed)
      // st[kt][r]: raw score of key k0 + 16kt + 4g + r against query qrow
      if (BIAS) {                                      // scores are scaled later by c = scale * log2(e): add bias / scale here
        const float inv_scale = 1.f / p.scale;
        const int qc = min(qrow, p.Sq - 1);
        const float* bh = p.bias_h ? p.bias_h + ((long)h * p.Sq + qc) * p.Skv : nullptr;
        const float* bb = p.bias_b ? p.bias_b + ((long)(b % p.bias_nb) * p.Sq + qc) * p.Skv : nullptr;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = min(k0 + kt * 16 + 4 * g + r, p.Skv - 1);
            float add = 0.f;
            if (bh) add += bh[key];
            if (bb) add += bb[key];
            st[kt][r] = fmaf(add, inv_scale, st[kt][r]);
          }
      }
      const bool need_mask = (k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw0 + off)) || (p.window > 0);
      if (need_mask) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = k0 + kt * 16 + 4 * g + r;
            if (key >= kvlen || (CAUSAL && key > qrow + off) || (p.window > 0 && key <= qrow + off - p.window)) st[kt][r] = -INFINITY;
          }
      }
      // LANE-LOCAL max of this lane's 16 scores; the cross-lane row max (two LDS-crossbar shuffles, ~100 cycles of latency each) is only
      // formed when some lane sees a score above m + 2^8: with every score <= m + RESCALE_THR the exponentials stay <= 2^8, so m may lag the
      // true running max (O / l does not depend on m).  The row sum l is kept per lane and folded across lanes once, after the loop.
      float mx;
      {
        const float m0 = vmax3(st[0][0], st[0][1], st[0][2]), m1 = vmax3(st[0][3], st[1][0], st[1][1]), m2 = vmax3(st[1][2], st[1][3], st[2][0]);
        const float m3 = vmax3(st[2][1], st[2][2], st[2][3]), m4 = vmax3(st[3][0], st[3][1], st[3][2]);
        mx = vmax3(vmax3(m0, m1, st[3][3]), vmax3(m2, m3, m4), -INFINITY);
      }
      mx *= c;
      if (!__all(mx <= m + RESCALE_THR)) {            // rare after the first tiles: rescale everything held at the old max
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mnew);
        l *= alpha;
#pragma unroll
        for (int d = 0; d < NDB; ++d) oacc[d] *= alpha;
        m = mnew;
      }
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp2(fmaf(st[kt][r], c, -m));
          st[kt][r] = e;
          rs += e;
        }
      l += rs;
      ATTN_STAMP(2)                                    // mask + softmax
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 pf = pack8(st[2 * kk], st[2 * kk + 1]);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const bf16x8 vf = trfrag(Vs, LD, kk * 32, d * 16, lane);
          oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (DBG) {
#pragma unroll
        for (int d = 0; d < NDB; ++d) asm volatile("" :: "v"(oacc[d]));
        asm volatile("s_nop 0" ::: "memory");
      }
      ATTN_STAMP(3)                                    // O += V^T P done
    }
    if (more) {                       // other buffer: last read one iteration ago, i.e. before the previous barrier
      kr.store(Kbuf + (cur ^ 1) * TILE, LD);
      vr.store(Vbuf + (cur ^ 1) * TILE, LD);
    }
    if (DBG) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ATTN_STAMP(4)                                      // wait for the loaded tile + LDS stores
    __syncthreads();
    ATTN_STAMP(5)                                      // barrier
    if (DBG && dbg_on) dbg_acc[7] += 1;
  }
  if (DBG && dbg_on) {
#pragma unroll
    for (int i = 0; i < 8; ++i) vp_attn_dbg[(wave == 7) * 8 + i] = dbg_acc[i];
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
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
  // D/8 lanes per row (16-byte loads), 64/(D/8)-ish rows per wave pass; rows = (b, h, q) flattened with q fastest
  constexpr int LPR = D / 8 <= 4 ? 4 : (D / 8 <= 8 ? 8 : 16);          // lanes per row (power of two >= D/8)
  constexpr int RPW = 64 / LPR;
  const long rows = (long)p.B * p.Hq * p.Sq;
  const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR;
  for (long i0 = (blockIdx.x * 4L + (threadIdx.x >> 6)) * RPW; i0 < rows; i0 += gridDim.x * 4L * RPW) {
    const long i = min(i0 + rsel, rows - 1);
    const int q = (int)(i % p.Sq);
    const int h = (int)((i / p.Sq) % p.Hq);
    const int b = (int)(i / ((long)p.Sq * p.Hq));
    float a = 0.f;
    if (sub * 8 < D) {
      const bf16x8 o = *(const bf16x8*)(p.o + (long)b * p.o_bs + (long)q * p.o_ts + (long)h * D + sub * 8);
      const bf16x8 d = *(const bf16x8*)(p.dout + (long)b * p.do_bs + (long)q * p.do_ts + (long)h * D + sub * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a += bf2f((bf16_t)o[e]) * bf2f((bf16_t)d[e]);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (sub == 0 && i0 + rsel < rows) {
      p.delta[i] = a;
      *(float2*)(p.delta + rows + 2 * i) = float2{p.lse[i], a};      // (lse, delta) pairs for the DMA-fed dK/dV kernel
    }
  }
}

// ================================================================================================
// backward: dK, dV  (block = 128 keys of one kv head, 16 per wave; loops the GQA group's q heads and 32-query tiles)
// ================================================================================================
template <int D, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_kernel(AttnParams p) {
  constexpr int LD = D + 16, NKS = D / 32, NDB = D / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Qbuf[2 * 32 * LD];
  __shared__ __attribute__((aligned(16))) bf16_t dObuf[2 * 32 * LD];
  __shared__ float lse_buf[2 * 32], delta_buf[2 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int hk = blockIdx.x, b = VP_BY(p);           // z (slowest dispatch index) = key block: early keys (most queries) first
  const int k0 = VP_BZ(p) * 128, kw0 = k0 + wave * 16, key = kw0 + fr;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const int rep = p.Hq / p.Hkv;
  const float c = p.scale * LOG2E;

  bf16x8 kf[NKS], vf[NKS];
  {
    const int keyc = min(key, p.Skv - 1);                   // clamped; keys >= kv_len are masked (p = 0) and not stored
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)keyc * p.k_ts + (long)hk * D;
    const bf16_t* vp = p.v + (long)b * p.v_bs + (long)keyc * p.v_ts + (long)hk * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[ks] = *(const bf16x8*)(kp + ks * 32 + g * 8);
      vf[ks] = *(const bf16x8*)(vp + ks * 32 + g * 8);
    }
  }
  f32x4 dk[NDB], dv[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  int qstart = 0, qend = p.Sq;
  if (CAUSAL) qstart = max(0, k0 - off) & ~31;
  if (p.window > 0) qend = min(p.Sq, k0 + 128 - off + p.window);
  if (k0 >= kvlen) qend = qstart;                      // whole key tile is padding: gradients are zero
  const int ntq = qend > qstart ? (qend - qstart + 31) / 32 : 0;
  const int nit = ntq * rep;                           // flattened (head, q tile) iteration space

  TileRegs<D, 32> qr, dor;
  float lse_r = 0.f, del_r = 0.f;
  auto prefetch = [&](int it) {
    const int h = hk * rep + it / ntq, q0 = qstart + (it % ntq) * 32;
    qr.load(p.q + (long)b * p.q_bs + (long)h * D, p.q_ts, q0, p.Sq);
    dor.load(p.dout + (long)b * p.do_bs + (long)h * D, p.do_ts, q0, p.Sq);
    const int qq = min(q0 + (int)(threadIdx.x & 31), p.Sq - 1);
    const long si = ((long)b * p.Hq + h) * p.Sq + qq;
    lse_r = p.lse[si];
    del_r = p.delta[si];
  };
  auto commit = [&](int buf) {
    qr.store(Qbuf + buf * 32 * LD, LD);
    dor.store(dObuf + buf * 32 * LD, LD);
    if (threadIdx.x < 32) { lse_buf[buf * 32 + threadIdx.x] = lse_r; delta_buf[buf * 32 + threadIdx.x] = del_r; }
  };
  if (nit > 0) { prefetch(0); commit(0); }
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    if (it + 1 < nit) prefetch(it + 1);
    const int cb = it & 1;
    const bf16_t* Qs = Qbuf + cb * 32 * LD;
    const bf16_t* dOs = dObuf + cb * 32 * LD;
    const float* lse_s = lse_buf + cb * 32;
    const float* delta_s = delta_buf + cb * 32;
    const int q0 = qstart + (it % ntq) * 32;
    // wave-uniform skip: every query of this tile is below this wave's first key (causal) -> all p = 0
    const bool active = !CAUSAL || (q0 + 31 + off >= kw0);
    if (active) {
      f32x4 s[2], dp[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        s[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(Qs + (qt * 16 + fr) * LD + ks * 32 + g * 8);
          const bf16x8 da = *(const bf16x8*)(dOs + (qt * 16 + fr) * LD + ks * 32 + g * 8);
          s[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], s[qt], 0, 0, 0);
          dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dp[qt], 0, 0, 0);
        }
      }
      // s[qt][r]: query = q0 + 16qt + 4g + r, key = this lane's key
      const bool need_mask = (q0 + 32 > p.Sq) || (kw0 + 16 > kvlen) || (CAUSAL && (kw0 + 15 > q0 + off)) || (p.window > 0);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qt * 16 + 4 * g + r;
          float pv = fast_exp2(fmaf(s[qt][r], c, -lse_s[ql]));
          if (need_mask) {
            const int qg = q0 + ql;
            const bool ok = qg < p.Sq && key < kvlen && (!CAUSAL || key <= qg + off) && (p.window <= 0 || key > qg + off - p.window);
            pv = ok ? pv : 0.f;
          }
          s[qt][r] = pv;
          dp[qt][r] = pv * (dp[qt][r] - delta_s[ql]);
        }
      const bf16x8 pf = pack8(s[0], s[1]);
      const bf16x8 dsf = pack8(dp[0], dp[1]);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const bf16x8 ta = trfrag(dOs, LD, 0, d * 16, lane);
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, pf, dv[d], 0, 0, 0);
        const bf16x8 tq = trfrag(Qs, LD, 0, d * 16, lane);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq, dsf, dk[d], 0, 0, 0);
      }
    }
    if (it + 1 < nit) commit(cb ^ 1);      // other buffer: last read one iteration ago (before the previous barrier)
    __syncthreads();
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
// backward: dQ  (block = 128 queries of one q head, 16 per wave; loops 64-key tiles)
// ================================================================================================
template <int D, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr int LD = D + 16, NKS = D / 32, NDB = D / 16, TILE = 64 * LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  bf16_t* const Kbuf = (bf16_t*)attn_smem;
  bf16_t* const Vbuf = Kbuf + 2 * TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nqb = (p.Sq + 127) >> 7;
  const int qb = nqb - 1 - VP_BZ(p);
  // blocks are dealt round-robin to the 8 XCDs (x & 7): give each XCD whole GQA groups so K/V tiles are shared in its L2
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 128, qw0 = q0 + wave * 16, qrow = qw0 + fr;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;

  bf16x8 qf[NKS], dof[NKS];
  const int qrc = min(qrow, p.Sq - 1);                        // clamped (unconditional loads)
  {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)qrc * p.q_ts + (long)h * D;
    const bf16_t* dp_ = p.dout + (long)b * p.do_bs + (long)qrc * p.do_ts + (long)h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *(const bf16x8*)(qp + ks * 32 + g * 8);
      dof[ks] = *(const bf16x8*)(dp_ + ks * 32 + g * 8);
    }
  }
  const long sidx = ((long)b * p.Hq + h) * p.Sq + qrc;
  const float lse = p.lse[sidx], dlt = p.delta[sidx];
  f32x4 dq[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 128 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~63;
  const bf16_t* kbase = p.k + (long)b * p.k_bs + (long)hk * D;
  const bf16_t* vbase = p.v + (long)b * p.v_bs + (long)hk * D;

  TileRegs<D, 64> kr, vr;
  if (kstart < kend) {
    kr.load(kbase, p.k_ts, kstart, p.Skv);
    vr.load(vbase, p.v_ts, kstart, p.Skv);
    kr.store(Kbuf, LD);
    vr.store(Vbuf, LD);
  }
  __syncthreads();
  int cur = 0;
  for (int k0 = kstart; k0 < kend; k0 += 64, cur ^= 1) {
    const bool more = k0 + 64 < kend;
    if (more) {
      kr.load(kbase, p.k_ts, k0 + 64, p.Skv);
      vr.load(vbase, p.v_ts, k0 + 64, p.Skv);
    }
    const bf16_t* Ks = Kbuf + cur * TILE;
    const bf16_t* Vs = Vbuf + cur * TILE;
    const bool active = !CAUSAL || (k0 <= qw0 + 15 + off);
    if (active) {
      const bool need_mask = (qw0 + 16 > p.Sq) || (k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw0 + off)) || (p.window > 0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f32x4 st[2], dpt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kk + t;
          st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          dpt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 ka = *(const bf16x8*)(Ks + (kt * 16 + fr) * LD + ks * 32 + g * 8);
            const bf16x8 va = *(const bf16x8*)(Vs + (kt * 16 + fr) * LD + ks * 32 + g * 8);
            st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ks], st[t], 0, 0, 0);
            dpt[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[ks], dpt[t], 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = fast_exp2(fmaf(st[t][r], c, -lse));
            if (need_mask) {
              const int key = k0 + (2 * kk + t) * 16 + 4 * g + r;
              const bool ok = qrow < p.Sq && key < kvlen && (!CAUSAL || key <= qrow + off) && (p.window <= 0 || key > qrow + off - p.window);
              pv = ok ? pv : 0.f;
            }
            dpt[t][r] = pv * (dpt[t][r] - dlt);
          }
        const bf16x8 dsf = pack8(dpt[0], dpt[1]);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const bf16x8 ktf = trfrag(Ks, LD, kk * 32, d * 16, lane);
          dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[d], 0, 0, 0);
        }
      }
    }
    if (more) {
      kr.store(Kbuf + (cur ^ 1) * TILE, LD);
      vr.store(Vbuf + (cur ^ 1) * TILE, LD);
    }
    __syncthreads();
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

#include "attention_d128.h"
#include "attention_bwd64.h"

// ================================================================================================
// C ABI
// ================================================================================================
template <int D>
static constexpr int kv_lds_bytes() { return 4 * 64 * (D + 16) * 2; }     // K,V tiles x 2 buffers

static int vp_attn_order() {                           // VP_ATTN_ORDER=0|1 (see AttnParams::order)
  static int v = -1;
  if (v < 0) { const char* e = getenv("VP_ATTN_ORDER"); v = e ? atoi(e) : 0; }
  return v;
}

static bool vp_fwdm_enabled() {                        // the 32x32x16 swapped-product forward (D = 128) is the default since round 3; VP_ATTN_FWDM=0: the 16-row kernel
  static int v = -1;
  if (v < 0) { const char* e = getenv("VP_ATTN_FWDM"); v = e ? atoi(e) : 1; }
  return v != 0;
}

template <int D>
static int launch_fwd(const AttnParams& p_in, int causal, hipStream_t s) {
  AttnParams p = p_in;
  p.order = (p.B <= 65535 && (p.Sq + 127) / 128 <= 65535) ? vp_attn_order() : 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<D, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<D, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    attr = true;
  }
  const dim3 grid = p.order ? dim3(p.Hq, (p.Sq + 127) / 128, p.B) : dim3(p.Hq, p.B, (p.Sq + 127) / 128);
  if constexpr (D == 128 || D == 96) {
    if (!p.bias_h && !p.bias_b && vp_fwdm_enabled()) {                  // 32x32x16 swapped-product kernel, 256-row blocks (attention_d128.h)
      static bool attrm = false;
      if (!attrm) {
        (void)hipFuncSetAttribute((const void*)attn_fwd128m_kernel<true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, FWDM_LDS);
        (void)hipFuncSetAttribute((const void*)attn_fwd128m_kernel<false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, FWDM_LDS);
        attrm = true;
      }
      const int nb = (p.Sq + 255) / 256;
      const dim3 gm = p.order ? dim3(p.Hq, nb, p.B) : dim3(p.Hq, p.B, nb);
      if (causal) hipLaunchKernelGGL((attn_fwd128m_kernel<true, D>), gm, dim3(512), FWDM_LDS, s, p);
      else hipLaunchKernelGGL((attn_fwd128m_kernel<false, D>), gm, dim3(512), FWDM_LDS, s, p);
      return vp_check_launch("vp_attn_fwd");
    }
  }
  if (D == 128 && causal && !p.bias_h && !p.bias_b && getenv("VP_ATTN_DBG")) {       // dev aid: phase stamps
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<D, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    hipLaunchKernelGGL((attn_fwd_kernel<D, true, false, true>), grid, dim3(512), kv_lds_bytes<D>(), s, p);
    return vp_check_launch("vp_attn_fwd");
  }
  if (p.bias_h || p.bias_b) {
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<D, true, true>), grid, dim3(512), kv_lds_bytes<D>(), s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<D, false, true>), grid, dim3(512), kv_lds_bytes<D>(), s, p);
  } else {
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<D, true>), grid, dim3(512), kv_lds_bytes<D>(), s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<D, false>), grid, dim3(512), kv_lds_bytes<D>(), s, p);
  }
  return vp_check_launch("vp_attn_fwd");
}
// VP_ATTN_BWD64: 1 (default) = the one-wave-per-SIMD backward kernels of round 5 (attention_bwd64.h); 2 = round 5's dK/dV kernel behind round 4's
// dQ kernel (which then writes the statistics planes); 0 = round 4's kernels
static int vp_bwd64_mode() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("VP_ATTN_BWD64"); v = e ? atoi(e) : 1; }
  return v;
}

template <int D>
static int launch_bwd(const AttnParams& p_in, int causal, hipStream_t s) {
  AttnParams p = p_in;
  p.order = (p.B <= 65535 && (p.Skv + 127) / 128 <= 65535 && (p.Sq + 127) / 128 <= 65535) ? vp_attn_order() : 0;
  const dim3 g1 = p.order ? dim3(p.Hkv, (p.Skv + 127) / 128, p.B) : dim3(p.Hkv, p.B, (p.Skv + 127) / 128);
  const dim3 g2 = p.order ? dim3(p.Hq, (p.Sq + 127) / 128, p.B) : dim3(p.Hq, p.B, (p.Sq + 127) / 128);
  if constexpr (D == 128 || D == 96) {
    if (vp_bwd64_mode() != 0) {                         // round 5: one wave per SIMD, 64 rows per wave, 32x32x16 MFMAs (attention_bwd64.h)
      if (p.rope_cos && !causal) { vp_set_error("vp_attn_bwd_rope: causal attention only"); return VP_ERR_UNSUPPORTED_SHAPE; }
      const int nkb = (p.Skv + 255) / 256, nqb = (p.Sq + 255) / 256;
      const dim3 h1 = p.order ? dim3(p.Hkv, nkb, p.B) : dim3(p.Hkv, p.B, nkb);
      const dim3 h2 = p.order ? dim3(p.Hq, nqb, p.B) : dim3(p.Hq, p.B, nqb);
      const bool old_dq = vp_bwd64_mode() == 2;
      if (old_dq) p.stat_planes = 1;
      static bool attr_o = false;
      if (old_dq && !attr_o) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<true, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<false, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<true, true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
        attr_o = true;
      }
#define VP_B64_LAUNCH(C_, R_)                                                                                                        \
  {                                                                                                                                  \
    static bool a_ = false;                                                                                                          \
    if (!a_) {                                                                                                                       \
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq64w_kernel<C_, R_, D>, hipFuncAttributeMaxDynamicSharedMemorySize, B64_DQ_LDS);   \
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv64w_kernel<C_, R_, D>, hipFuncAttributeMaxDynamicSharedMemorySize, B64_KV_LDS); \
      a_ = true;                                                                                                                     \
    }                                                                                                                                \
    if (old_dq) hipLaunchKernelGGL((attn_bwd_dq128_kernel<C_, R_, D>), g2, dim3(256), DQ128_LDS, s, p);                              \
    else hipLaunchKernelGGL((attn_bwd_dq64w_kernel<C_, R_, D>), h2, dim3(256), B64_DQ_LDS, s, p);     /* first: writes the statistics planes */ \
    hipLaunchKernelGGL((attn_bwd_dkdv64w_kernel<C_, R_, D>), h1, dim3(256), B64_KV_LDS, s, p);                                       \
  }
      if (p.rope_cos) VP_B64_LAUNCH(true, true)
      else if (causal) VP_B64_LAUNCH(true, false)
      else VP_B64_LAUNCH(false, false)
#undef VP_B64_LAUNCH
      return vp_check_launch("vp_attn_bwd");
    }
    // the LDS-DMA ring kernels (attention_d128.h): dQ first — it also writes the (lse, delta) pairs the dK/dV kernel streams
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv128_kernel<true, DKDV_KT, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DKDV128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv128_kernel<false, DKDV_KT, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DKDV128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<true, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<false, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
      attr2 = true;
    }
    if (p.rope_cos) {                                   // fused RoPE backward: its own instantiations (training: causal only)
      if (!causal) { vp_set_error("vp_attn_bwd_rope: causal attention only"); return VP_ERR_UNSUPPORTED_SHAPE; }
      static bool attr3 = false;
      if (!attr3) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv128_kernel<true, DKDV_KT, true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DKDV128_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<true, true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
        attr3 = true;
      }
      hipLaunchKernelGGL((attn_bwd_dq128_kernel<true, true, D>), g2, dim3(256), DQ128_LDS, s, p);
      hipLaunchKernelGGL((attn_bwd_dkdv128_kernel<true, DKDV_KT, true, D>), g1, dim3(64 * (8 / DKDV_KT)), DKDV128_LDS, s, p);
    } else if (causal) {
      hipLaunchKernelGGL((attn_bwd_dq128_kernel<true, false, D>), g2, dim3(256), DQ128_LDS, s, p);
      hipLaunchKernelGGL((attn_bwd_dkdv128_kernel<true, DKDV_KT, false, D>), g1, dim3(64 * (8 / DKDV_KT)), DKDV128_LDS, s, p);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq128_kernel<false, false, D>), g2, dim3(256), DQ128_LDS, s, p);
      hipLaunchKernelGGL((attn_bwd_dkdv128_kernel<false, DKDV_KT, false, D>), g1, dim3(64 * (8 / DKDV_KT)), DKDV128_LDS, s, p);
    }
  } else {
    const long rows = (long)p.B * p.Hq * p.Sq;
    hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)min(8192L, (rows + 15) / 16)), dim3(256), 0, s, p);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
      attr = true;
    }
    if (causal) {
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, true>), g1, dim3(512), 0, s, p);
      hipLaunchKernelGGL((attn_bwd_dq_kernel<D, true>), g2, dim3(512), kv_lds_bytes<D>(), s, p);
    } else {
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, false>), g1, dim3(512), 0, s, p);
      hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false>), g2, dim3(512), kv_lds_bytes<D>(), s, p);
    }
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

// dev aid: the forward kernel's phase stamps of the last VP_ATTN_DBG launch (16 longs: waves 0 and 7 of block (0,0,0))
#ifdef VP_DEBUG
int vp_debug_attn_stamps(long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(vp_attn_dbg), sizeof(long) * 16) == hipSuccess ? 0 : 1; }
#endif

// Same, with additive fp32 score biases (Swin window attention, HF modeling_swin.py SwinAttention.forward: relative position bias per
// head [Hq,Sq,Skv] + shifted-window mask [bias_nb,Sq,Skv] indexed by batch % bias_nb).  Forward only (frozen teacher).
int vp_attn_fwd_bias(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k, long k_bs,
                long k_ts, const void* v, long v_bs, long v_ts, void* o, long o_bs, long o_ts, float* lse, const int* kv_len,
                int causal, int window, float scale, const float* bias_h, const float* bias_b, int bias_nb,
                     hipStream_t s) {
  int e = check_attn("vp_attn_fwd_bias", B, Hq, Hkv, Sq, Skv, D);
  if (e) return e;
  AttnParams p{};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
  p.q_bs = q_bs; p.q_ts = q_ts; p.k_bs = k_bs; p.k_ts = k_ts; p.v_bs = v_bs; p.v_ts = v_ts; p.o_bs = o_bs; p.o_ts = o_ts;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv; p.window = window; p.kv_len = kv_len; p.scale = scale;
  VP_REQUIRE(!bias_b || bias_nb > 0, VP_ERR_BAD_ARG, "vp_attn_fwd_bias: bias_nb");
  p.bias_h = bias_h; p.bias_b = bias_b; p.bias_nb = bias_nb;
  switch (D) {
    case 32: return launch_fwd<32>(p, causal, s);
    case 64: return launch_fwd<64>(p, causal, s);
    case 96: return launch_fwd<96>(p, causal, s);
    default: return launch_fwd<128>(p, causal, s);
  }
}

// delta: fp32 workspace of 3 * B * Hq * Sq floats (delta, then interleaved (lse, delta) pairs).  dq/dk/dv use the same [B,S,H,D] addressing with their own strides.
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

// vp_attn_bwd with the RoPE backward of dq / dk fused into the stores (reference: HF LlamaAttention.forward applies apply_rotary_pos_emb
// to q, k before SDPA -- modeling_llama.py / modeling_phi3.py; its autograd rotates dq / dk back).  D = 128 or 96 (Phi-3), causal only.
// rope_cos / rope_sin: fp32 [positions, D / 2] as for vp_rope; rope_pos: int32 [B, S] position ids or NULL (position = row index).  Same numbers as vp_attn_bwd
// followed by vp_rope(inverse = 1) on dq and dk.
int vp_attn_bwd_rope(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k, long k_bs,
                     long k_ts, const void* v, long v_bs, long v_ts, const void* o, long o_bs, long o_ts, const float* lse,
                     const void* dout, long do_bs, long do_ts, void* dq, long dq_bs, long dq_ts, void* dk, long dk_bs, long dk_ts,
                     void* dv, long dv_bs, long dv_ts, float* delta, const int* kv_len, int causal, int window, float scale,
                     const float* rope_cos, const float* rope_sin, const int* rope_pos, hipStream_t s) {
  int e = check_attn("vp_attn_bwd_rope", B, Hq, Hkv, Sq, Skv, D);
  if (e) return e;
  VP_REQUIRE(lse && delta && dout && dq && dk && dv && rope_cos && rope_sin, VP_ERR_BAD_ARG, "vp_attn_bwd_rope: null pointer");
  VP_REQUIRE((D == 128 || D == 96) && causal, VP_ERR_UNSUPPORTED_SHAPE, "vp_attn_bwd_rope: head_dim 128 or 96, causal only (got %d, causal %d)", D, causal);
  VP_REQUIRE(((((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15) == 0, VP_ERR_BAD_ARG, "vp_attn_bwd_rope: tables must be 16-byte aligned");
  AttnParams p{};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = (float*)lse;
  p.dout = (const bf16_t*)dout; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.delta = delta;
  p.q_bs = q_bs; p.q_ts = q_ts; p.k_bs = k_bs; p.k_ts = k_ts; p.v_bs = v_bs; p.v_ts = v_ts; p.o_bs = o_bs; p.o_ts = o_ts;
  p.do_bs = do_bs; p.do_ts = do_ts; p.dq_bs = dq_bs; p.dq_ts = dq_ts; p.dk_bs = dk_bs; p.dk_ts = dk_ts; p.dv_bs = dv_bs; p.dv_ts = dv_ts;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv; p.window = window; p.kv_len = kv_len; p.scale = scale;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_pos = rope_pos;
  return D == 128 ? launch_bwd<128>(p, causal, s) : launch_bwd<96>(p, causal, s);
}

}  // extern "C"
