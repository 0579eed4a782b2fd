// Flash-style softmax attention for the VisPer-LM hot path (forward + backward), bf16 MFMA 16x16x32.
//   * Llama/Phi-3 decoder: causal, GQA, D=128/96            (fwd + bwd; LLM is frozen -> dgrad only)
//   * CLIP-ViT: non-causal, N=577, D=64                      (fwd only)
//   * Perceiver resampler heads: cross-attention, D=32       (fwd + bwd)
// Design (CDNA4, 64-lane waves):
//  - every kernel keeps the *query* (fwd, dQ) or *key* (dK/dV) index on the lane axis so softmax statistics
//    are lane-local, and builds the second GEMM's operand directly from the first GEMM's accumulator
//    registers (no LDS round trip for P / dS): a 16x16x32 MFMA contracts over 32 "k slots" (g = lane>>4,
//    j = 0..7); A and B only have to agree on which token each slot means, so slot (g,j) is mapped to token
//    16*(j>>2) + 4g + (j&3) — exactly the (row = 4g + r) layout two stacked C tiles have.
//  - all LDS tiles are plain row-major [token][feature] (stride D+16 elements: conflict-free for both read
//    kinds).  Fragments contracted over features are ds_read_b128; fragments contracted over TOKENS (V in
//    fwd, Q/dO in dK/dV, K in dQ) come from the gfx950 hardware transpose read ds_read_b64_tr_b16
//    (lane 4a+b supplies the address of row a, features 4b..4b+3; it receives column (lane&15) of that
//    4x16 block), so nothing is ever staged transposed.
//  - the next K/V (or Q/dO) tile is prefetched HBM->registers while the current tile is being consumed and written
//    into the OTHER LDS buffer after the compute (issue-early / write-late, double-buffered LDS): HBM latency hides
//    under MFMA and there is ONE barrier per tile.
//  - 8 waves (512 threads) x 16 rows per block at <= 128 VGPRs: the kernels are VALU-issue bound (softmax /
//    dS arithmetic ~ as many issue cycles as the MFMAs), so they want 4 waves per SIMD to hide dependent-issue
//    latency more than they want bigger per-wave tiles (LDS is only ~10 % busy).  exp2 is the raw v_exp_f32
//    with scale*log2(e) folded into one FMA; the O rescale is skipped while the running max grows < 2^8.
// Log-sum-exp is kept in the log2 domain: lse2 = m + log2(l) with scores pre-multiplied by scale*log2(e).
#include "common.h"
#include <stdlib.h>

struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  const bf16_t* dout; bf16_t* dq; bf16_t* dk; bf16_t* dv; float* delta;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  long do_bs, do_ts, dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
  int B, Hq, Hkv, Sq, Skv, window;
  const int* kv_len;
  float scale;
  // optional additive score biases (forward only; Swin window attention): bias_h [Hq, Sq, Skv] per head (relative position bias),
  // bias_b [bias_nb, Sq, Skv] indexed by batch % bias_nb (the shifted-window mask); fp32, added to the scaled scores
  const float* bias_h;
  const float* bias_b;
  int bias_nb;
  // optional fused RoPE backward (D = 128 DMA-ring kernels only): dq / dk leave the kernels already rotated back (HF apply_rotary_pos_emb
  // autograd, same rounding points as vp_rope(inverse=1) applied to the bf16 dq / dk).  cos / sin fp32 [positions, 64]; pos int [B, S] or NULL
  const float* rope_cos;
  const float* rope_sin;
  const int* rope_pos;
  // dispatch order of the 3-D grid: 0 = (head, batch, tile) with `tile` slowest (all batches of one tile level run together);
  // 1 = (head, tile, batch) with `batch` slowest: one batch's heads and tiles run together, so an XCD's resident blocks share ONE
  // (batch, kv-head) K/V set (1 MB at S=2048) instead of eight (8 MB > the 4 MB L2)
  int order;
};
#define VP_BY(P) ((P).order ? (int)blockIdx.z : (int)blockIdx.y)      /* batch index */
#define VP_BZ(P) ((P).order ? (int)blockIdx.y : (int)blockIdx.z)      /* tile index */

#define LOG2E 1.4426950408889634f
#define RESCALE_THR 8.0f     // log2 units: skip the O/l rescale while the running max grows by < 2^8 (wave-uniform)
static __device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // raw v_exp_f32
typedef __attribute__((ext_vector_type(4))) short s16x4;
// 3-input max as ONE instruction: through fmaxf hipcc first canonicalises every MFMA output with a v_max_f32 x, x (NaN semantics are on:
// -fno-finite-math-only), 16 extra VALU instructions per 16-score tile; scores are finite or -inf here, so the raw instruction is exact
static __device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
static __device__ __forceinline__ bf16x8 zero8() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// 8 fp32 (two C tiles' registers) -> one bf16x8 MFMA operand
static __device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
  r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
  return r;
}

// Token-contracted fragment from a ROW-MAJOR tile t[token][feature] (stride ld): this lane (fr = lane&15,
// g = lane>>4) receives feature f0+fr of tokens tok0+4g+{0..3} and tok0+16+4g+{0..3}  (== k slots (g, 0..7)).
static __device__ __forceinline__ bf16x8 trfrag(const bf16_t* t, int ld, int tok0, int f0, int lane) {
  const int g = lane >> 4, a = (lane & 15) >> 2, b = lane & 3;
  const bf16_t* p0 = t + (tok0 + 4 * g + a) * ld + f0 + 4 * b;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * ld));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// R token rows x D features prefetched into registers (HBM latency hides under the current tile's MFMAs),
// then written to the row-major LDS tile.  Loads are UNCONDITIONAL (row index clamped to limit-1): a
// predicated load makes hipcc branch around it and wait vmcnt(0) per element, serialising the L2 round
// trips.  Out-of-range rows therefore hold a copy of the last valid row; every consumer masks them
// (scores of keys >= kv_len / queries >= Sq are forced to p = 0) so they never reach an output.
template <int D, int R, int NT = 512>
struct TileRegs {
  static constexpr int CH = D / 8;
  static constexpr int N = (R * CH + NT - 1) / NT;
  bf16x8 v[N];
  __device__ __forceinline__ void load(const bf16_t* gbase, long ts, int row0, int limit) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int it = threadIdx.x + i * NT;
      const int itc = (R * CH) % NT == 0 ? it : min(it, R * CH - 1);
      const int r = itc / CH, c = itc % CH;
      v[i] = *(const bf16x8*)(gbase + (long)min(row0 + r, limit - 1) * ts + c * 8);
    }
  }
  // Same tile through buffer loads: wave-uniform descriptor (base of this batch / head, num_records = bytes up to the end of row limit-1)
  // + a per-lane byte offset that never changes (voff(ts)) + the tile's row offset as the scalar offset: NO address arithmetic per tile
  // (the 64-bit multiply-adds and row clamps of load() were ~25 of the forward's ~144 VALU instructions per tile), and rows >= limit come
  // back as zeros from the range check instead of a clamped copy (every consumer masks them anyway).
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const bf16_t* gbase, long ts, int limit) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, (int)(((long)(limit - 1) * ts + D) * 2), 0x00020000);
  }
  __device__ __forceinline__ void voff(int ts_elems, int (&off)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int it = threadIdx.x + i * NT;
      const int itc = (R * CH) % NT == 0 ? it : min(it, R * CH - 1);
      off[i] = ((itc / CH) * ts_elems + (itc % CH) * 8) * 2;
    }
  }
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rs, const int (&off)[N], int row0, int ts_elems) {
    const int soff = row0 * ts_elems * 2;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], soff, 0));
  }
  __device__ __forceinline__ void store(bf16_t* lds, int ld) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int it = threadIdx.x + i * NT;
      const int r = it / CH, c = it % CH;
      if (it < R * CH) *(bf16x8*)(lds + r * ld + c * 8) = v[i];
    }
  }
};

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
// backward: dK, dV for D = 128 (the decoder's shape).  Block = 4 waves x 32 keys of one kv head; loops the GQA group's
// q heads and 32-query tiles.  The 16-keys-per-wave kernel above is HBM-*latency* bound (one 16 KB tile in flight per CU,
// ~3.7 us per iteration) and reads every Q/dO tile from LDS once per 16 keys.  Here:
//   * Q / dO / (lse, delta) tiles stream HBM -> LDS with global_load_lds into a 4-stage ring (3 tiles in flight, no staging
//     registers, one raw s_barrier per tile, counted vmcnt so the queue never drains);
//   * LDS tiles are dense [32][128] (DMA writes are lane-linear, no padding possible): the 16-byte chunk index is XOR-ed
//     with (row & 15) on the SOURCE side, which keeps both the row-wise ds_read_b128 and the transposing
//     ds_read_b64_tr_b16 fragment reads conflict-free;
//   * each wave owns 32 keys (two 16-key MFMA column tiles), so every LDS fragment feeds two MFMAs;
//   * 256 threads at <= 256 VGPRs -> two blocks per CU run out of phase.  (KT = 1, i.e. 8 waves x 16 keys at 128 VGPRs / 4 waves per
//     SIMD, was measured too: 1.46 ms against 0.74 ms -- twice the LDS traffic per MFMA and spills that drain the DMA ring.)
// ================================================================================================
#define ATTN_GLDS(gptr, ldsptr, BYTES)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),           \
                                   (__attribute__((address_space(3))) void*)(ldsptr), BYTES, 0, 0)
static __device__ __forceinline__ bf16x8 rowfrag_sw(const bf16_t* t, int row, int chunk) {
  return *(const bf16x8*)(t + row * 128 + ((chunk ^ (row & 15)) << 3));
}
static __device__ __forceinline__ bf16x8 trfrag_sw(const bf16_t* t, int f0, int lane) {
  const int g = lane >> 4, a = (lane & 15) >> 2, b = lane & 3;
  const int row = 4 * g + a, chunk = (f0 >> 3) + (b >> 1);
  const bf16_t* p0 = t + row * 128 + ((chunk ^ row) << 3) + (b & 1) * 4;     // rows +16 share the swizzle phase
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * 128));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// Transposing LDS reads of the DMA-ring kernels go through inline asm: via the builtin the compiler assumes the read may alias the
// in-flight global_load_lds writes and drains the whole ring (`s_waitcnt vmcnt(0)`) in front of every batch.  The kernels order ring
// stages themselves (counted vmcnt + s_barrier), and wait for these reads with ATTN_LGKM before pinning / using the result.
template <int OFF>
static __device__ __forceinline__ s16x4 tr_read_asm(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
static __device__ __forceinline__ uint32_t attn_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
static __device__ __forceinline__ bf16x8 tr_join(s16x4 lo, s16x4 hi) {
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
#define ATTN_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define ATTN_PIN(F) asm volatile("" : "+v"(F))

// store one row's 128 features (this lane: blocks d = 0..7, features d*16 + 4g + r) as bf16, optionally rotated back by RoPE^T:
// x1 = feature f < 64, x2 = feature f + 64; out1 = bf(bf(x1 c) + bf(x2 s)), out2 = bf(bf(x2 c) + bf(-x1 s))  (vp_rope with inverse = 1)
template <bool ROPE>
static __device__ __forceinline__ void store_row128(bf16_t* dst, const f32x4 (&acc)[8], float scale, int g, const float* cs, const float* sn) {
  if (!ROPE) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      bf16x4 a;
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = (short)f2bf(acc[d][r] * scale);
      *(bf16x4*)(dst + d * 16 + 4 * g) = a;
    }
  } else {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const f32x4 c4 = *(const f32x4*)(cs + d * 16 + 4 * g), s4 = *(const f32x4*)(sn + d * 16 + 4 * g);
      bf16x4 a, b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = bfround(acc[d][r] * scale), x2 = bfround(acc[d + 4][r] * scale);
        const float ss = -s4[r];
        a[r] = (short)f2bf(bfround(x1 * c4[r]) + bfround(-x2 * ss));
        b[r] = (short)f2bf(bfround(x2 * c4[r]) + bfround(x1 * ss));
      }
      *(bf16x4*)(dst + d * 16 + 4 * g) = a;
      *(bf16x4*)(dst + 64 + d * 16 + 4 * g) = b;
    }
  }
}
#ifndef DKDV_KT
#define DKDV_KT 2
#endif
constexpr int DKDV128_STAGE = 2 * 32 * 128 + 128;     // bf16 units: Q tile | dO tile | 32 (lse, delta) fp32 pairs
constexpr int DKDV128_LDS = 4 * DKDV128_STAGE * 2;    // bytes

template <bool CAUSAL, int KT, bool ROPE = false>      // KT = 16-key column tiles per wave: 2 -> 4 waves x 32 keys, 1 -> 8 waves x 16 keys (128 keys per block)
__global__ __launch_bounds__(64 * (8 / KT)) __attribute__((amdgpu_waves_per_eu(KT == 2 ? 2 : 4, KT == 2 ? 2 : 4)))
void attn_bwd_dkdv128_kernel(AttnParams p) {
  constexpr int D = 128, NKS = 4, NDB = 8, NW = 8 / KT, NI = 8 / NW;      // NI = DMA instructions per wave per 32-row tile
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  bf16_t* const ring = (bf16_t*)attn_smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, g = lane >> 4;
  const int hk = blockIdx.x, b = VP_BY(p);           // z (slowest dispatch index) = key block: early keys (most queries) first
  const int k0 = VP_BZ(p) * 128, kw0 = k0 + wave * 16 * KT;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const int rep = p.Hq / p.Hkv;
  const float c = p.scale * LOG2E;
  const float* pairs = p.delta + (long)p.B * p.Hq * p.Sq;            // (lse, delta) interleaved, written by the pre-pass

  bf16x8 kf[KT][NKS], vf[KT][NKS];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int keyc = min(kw0 + kt * 16 + fr, p.Skv - 1);    // clamped; keys >= kv_len are masked (p = 0) and not stored
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)keyc * p.k_ts + (long)hk * D;
    const bf16_t* vp = p.v + (long)b * p.v_bs + (long)keyc * p.v_ts + (long)hk * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[kt][ks] = *(const bf16x8*)(kp + ks * 32 + g * 8);
      vf[kt][ks] = *(const bf16x8*)(vp + ks * 32 + g * 8);
    }
  }
  f32x4 dk[KT][NDB], dv[KT][NDB];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int d = 0; d < NDB; ++d) { dk[kt][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  int qstart = 0, qend = p.Sq;
  if (CAUSAL) qstart = max(0, k0 - off) & ~31;
  if (p.window > 0) qend = min(p.Sq, k0 + 128 - off + p.window);
  if (k0 >= kvlen) qend = qstart;                      // whole key tile is padding: gradients are zero
  const int ntq = qend > qstart ? (qend - qstart + 31) / 32 : 0;
  const int nit = ntq * rep;                           // flattened (head, q tile) iteration space

  // DMA geometry: instruction i of this wave fills rows (i*4 + wave)*4 .. +3 of a tile; lane -> (row, 16-byte slot).
  // Rows r and r+16 share the swizzle phase, so both instructions use the same source chunk.
  // Everything derived from the lane id is RE-derived inside the loop behind an opaque asm: the register budget is full
  // (dk/dv 128 + K/V fragments 64), and hipcc would otherwise spill these loop invariants to scratch and reload them with
  // s_waitcnt vmcnt(0) -- which drains the DMA queue every iteration.
  auto issue = [&](int t, int st, int ln) {
    const int tc = min(t, nit - 1);                    // past the end: harmless re-fetch keeps the vmcnt bookkeeping uniform
    const int hh = tc / ntq;
    const int h = hk * rep + hh, q0 = qstart + (tc - hh * ntq) * 32;
    bf16_t* sb = ring + st * DKDV128_STAGE;
    const char* qb = (const char*)(p.q + (long)b * p.q_bs + (long)h * D);      // wave-uniform bases + 32-bit lane offsets
    const char* gb = (const char*)(p.dout + (long)b * p.do_bs + (long)h * D);
    const unsigned qts = (unsigned)p.q_ts, gts = (unsigned)p.do_ts;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int blk = i * NW + wave;                   // 4-row group of the tile this instruction fills
      const int drow = blk * 4 + (ln >> 4);
      const int dch = ((ln & 15) ^ (drow & 15)) * 8;
      const unsigned r = (unsigned)min(q0 + drow, p.Sq - 1);
      ATTN_GLDS(qb + (size_t)((r * qts + dch) * 2u), sb + blk * 512, 16);
      ATTN_GLDS(gb + (size_t)((r * gts + dch) * 2u), sb + 4096 + blk * 512, 16);
    }
    if (wave == 0) {
      const long qi = min(q0 + (ln >> 1), p.Sq - 1);
      ATTN_GLDS(pairs + (((long)b * p.Hq + h) * p.Sq + qi) * 2 + (ln & 1), sb + 8192, 4);
    }
  };
  if (nit > 0) { issue(0, 0, lane); issue(1, 1, lane); issue(2, 2, lane); }
  for (int it = 0; it < nit; ++it) {
    // tile `it` landed (this wave's part): at most the two younger stages may still be in flight
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (2 * NI + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NI) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                      // every wave's part landed; stage (it-1)&3 is no longer being read
    __builtin_amdgcn_sched_barrier(0);
    int ln = threadIdx.x & 63;
    asm volatile("" : "+v"(ln));                       // opaque: see the note above `issue`
    issue(it + 3, (it + 3) & 3, ln);
    const int fr = ln & 15, g = ln >> 4;
    // LDS fragment addressing: the swizzle is an XOR on the chunk bits, so one base per read kind + compile-time XOR masks
    const int rbase = fr * 128 + ((g ^ fr) << 3);                         // row-wise: (row fr, chunk g) ^ (ks*4 chunks), + qt*16 rows
    const int trow = 4 * g + (fr >> 2);
    const int tbase = trow * 128 + ((((ln & 3) >> 1) ^ trow) << 3) + (ln & 1) * 4;
    const bf16_t* Qs = ring + (it & 3) * DKDV128_STAGE;
    const bf16_t* dOs = Qs + 4096;
    const float* ld = (const float*)(Qs + 8192);
    const int hh = it / ntq;
    const int q0 = qstart + (it - hh * ntq) * 32;
    // wave-uniform skip: every query of this tile is below this wave's first key (causal) -> all p = 0
    const bool active = !CAUSAL || (q0 + 31 + off >= kw0);
    if (active) {
      // s[kt][r]: query = q0 + 16qt + 4g + r, key = kw0 + 16kt + fr.  One 16-query half at a time: only the packed bf16
      // P / dS halves stay live across the two halves (register budget: dk/dv 128 + K/V fragments 64).
      const bool need_mask = (q0 + 32 > p.Sq) || (kw0 + 16 * KT > kvlen) || (CAUSAL && (kw0 + 16 * KT - 1 > q0 + off)) || (p.window > 0);
      u32x2 pk[KT][2], dsk[KT][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4 s[KT], dp[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { s[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(Qs + (rbase ^ (ks * 32)) + qt * 2048);
          const bf16x8 da = *(const bf16x8*)(dOs + (rbase ^ (ks * 32)) + qt * 2048);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], s[kt], 0, 0, 0);
            dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[kt], 0, 0, 0);
          }
          if (ks & 1) __builtin_amdgcn_sched_barrier(0);     // bound the scheduler's look-ahead: fragments for <= 2 k-steps live
        }
        __builtin_amdgcn_s_setprio(0);
        const f32x4 l0 = *(const f32x4*)(ld + (qt * 16 + 4 * g) * 2), l1 = *(const f32x4*)(ld + (qt * 16 + 4 * g) * 2 + 4);
        const float lse_r[4] = {l0[0], l0[2], l1[0], l1[2]}, del_r[4] = {l0[1], l0[3], l1[1], l1[3]};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = fast_exp2(fmaf(s[kt][r], c, -lse_r[r]));
            if (need_mask) {
              const int qg = q0 + qt * 16 + 4 * g + r, key = kw0 + kt * 16 + fr;
              const bool ok = qg < p.Sq && key < kvlen && (!CAUSAL || key <= qg + off) && (p.window <= 0 || key > qg + off - p.window);
              pv = ok ? pv : 0.f;
            }
            s[kt][r] = pv;
            dp[kt][r] = pv * (dp[kt][r] - del_r[r]);
          }
          pk[kt][qt] = u32x2{pack_bf16x2(s[kt][0], s[kt][1]), pack_bf16x2(s[kt][2], s[kt][3])};
          dsk[kt][qt] = u32x2{pack_bf16x2(dp[kt][0], dp[kt][1]), pack_bf16x2(dp[kt][2], dp[kt][3])};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x8 pf[KT], dsf[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        pf[kt] = __builtin_bit_cast(bf16x8, u32x4{pk[kt][0][0], pk[kt][0][1], pk[kt][1][0], pk[kt][1][1]});
        dsf[kt] = __builtin_bit_cast(bf16x8, u32x4{dsk[kt][0][0], dsk[kt][0][1], dsk[kt][1][0], dsk[kt][1][1]});
      }
      const uint32_t qs_addr = attn_lds_addr(Qs);      // dO tile = +8192 bytes, rows +16 = +4096 bytes
      __builtin_amdgcn_s_setprio(1);                  // MFMA bursts at raised priority: the co-resident block's VALU/LDS work yields (-3 %)
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const uint32_t ta_ = qs_addr + 2u * (uint32_t)(tbase ^ (d * 16));
        const s16x4 al = tr_read_asm<8192>(ta_), ah = tr_read_asm<12288>(ta_);
        const s16x4 ql = tr_read_asm<0>(ta_), qh = tr_read_asm<4096>(ta_);
        ATTN_LGKM(2);
        bf16x8 ta = tr_join(al, ah);
        ATTN_PIN(ta);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) dv[kt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, pf[kt], dv[kt][d], 0, 0, 0);
        ATTN_LGKM(0);
        bf16x8 tq = tr_join(ql, qh);
        ATTN_PIN(tq);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) dk[kt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq, dsf[kt], dk[kt][d], 0, 0, 0);
        if (d & 1) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (dummy) DMAs must not outlive the block's LDS
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + kt * 16 + fr;
    if (key < p.Skv) {
      bf16_t* dkp = p.dk + (long)b * p.dk_bs + (long)key * p.dk_ts + (long)hk * D;
      bf16_t* dvp = p.dv + (long)b * p.dv_bs + (long)key * p.dv_ts + (long)hk * D;
      const long pp = ROPE ? (p.rope_pos ? (long)p.rope_pos[(long)b * p.Skv + key] : (long)key) * 64 : 0;
      store_row128<ROPE>(dkp, dk[kt], p.scale, g, p.rope_cos + pp, p.rope_sin + pp);
      store_row128<false>(dvp, dv[kt], 1.f, g, nullptr, nullptr);
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

// ================================================================================================
// backward: dQ for D = 128.  Same structure as the DMA-fed dK/dV kernel: block = 4 waves x 32 queries of one q head,
// K / V stream through a 4-stage LDS ring in 32-key tiles (global_load_lds, swizzled source chunks, counted vmcnt).
// ================================================================================================
constexpr int DQ128_STAGE = 2 * 32 * 128;             // bf16 units: K tile | V tile
constexpr int DQ128_LDS = 4 * DQ128_STAGE * 2;        // bytes

template <bool CAUSAL, bool ROPE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq128_kernel(AttnParams p) {
  constexpr int D = 128, NKS = 4, NDB = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  bf16_t* const ring = (bf16_t*)attn_smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (p.Sq + 127) >> 7;
  const int qb = nqb - 1 - VP_BZ(p);            // z is the slowest dispatch index: heavy (late) causal blocks first
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;     // whole GQA groups per XCD
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 128, qw0 = q0 + wave * 32;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;

  bf16x8 qf[2][NKS], dof[2][NKS];
  float lse[2], dlt[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrc = min(qw0 + qt * 16 + (lane & 15), p.Sq - 1);          // clamped (unconditional loads)
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)qrc * p.q_ts + (long)h * D;
    const bf16_t* dp_ = p.dout + (long)b * p.do_bs + (long)qrc * p.do_ts + (long)h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[qt][ks] = *(const bf16x8*)(qp + ks * 32 + (lane >> 4) * 8);
      dof[qt][ks] = *(const bf16x8*)(dp_ + ks * 32 + (lane >> 4) * 8);
    }
    const long sidx = ((long)b * p.Hq + h) * p.Sq + qrc;
    lse[qt] = p.lse[sidx];
    // delta = rowsum(dO * O) is computed here instead of in a pre-pass kernel (every (b, h, q) row belongs to exactly one wave of this
    // grid; the dO fragments are already in registers); the (lse, delta) pairs go to the workspace the dK/dV kernel streams from, so
    // this kernel is launched first
    const bf16_t* op_ = p.o + (long)b * p.o_bs + (long)qrc * p.o_ts + (long)h * D;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bf16x8 ov = *(const bf16x8*)(op_ + ks * 32 + (lane >> 4) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(bf2f((bf16_t)dof[qt][ks][e]), bf2f((bf16_t)ov[e]), part);
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    dlt[qt] = part;
    if ((lane >> 4) == 0 && qw0 + qt * 16 + (lane & 15) < p.Sq) {
      p.delta[sidx] = part;
      *(float2*)(p.delta + (long)p.B * p.Hq * p.Sq + 2 * sidx) = float2{lse[qt], part};
    }
  }
  f32x4 dq[2][NDB];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int d = 0; d < NDB; ++d) dq[qt][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 128 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~31;
  const int nit = kend > kstart ? (kend - kstart + 31) / 32 : 0;
  const char* kb = (const char*)(p.k + (long)b * p.k_bs + (long)hk * D);
  const char* vb = (const char*)(p.v + (long)b * p.v_bs + (long)hk * D);

  auto issue = [&](int t, int st, int ln) {            // lane-derived values are re-derived per call (see the dK/dV kernel)
    const int k0 = kstart + min(t, nit - 1) * 32;
    const int drow = wave * 4 + (ln >> 4);
    const int dch = ((ln & 15) ^ (drow & 15)) * 8;
    bf16_t* sb = ring + st * DQ128_STAGE;
    const unsigned r0 = (unsigned)min(k0 + drow, p.Skv - 1), r1 = (unsigned)min(k0 + drow + 16, p.Skv - 1);
    const unsigned kts = (unsigned)p.k_ts, vts = (unsigned)p.v_ts;
    ATTN_GLDS(kb + (size_t)((r0 * kts + dch) * 2u), sb + wave * 512, 16);
    ATTN_GLDS(kb + (size_t)((r1 * kts + dch) * 2u), sb + (4 + wave) * 512, 16);
    ATTN_GLDS(vb + (size_t)((r0 * vts + dch) * 2u), sb + 4096 + wave * 512, 16);
    ATTN_GLDS(vb + (size_t)((r1 * vts + dch) * 2u), sb + 4096 + (4 + wave) * 512, 16);
  };
  if (nit > 0) { issue(0, 0, lane); issue(1, 1, lane); issue(2, 2, lane); }
  for (int it = 0; it < nit; ++it) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile `it` landed (this wave's part); two younger stages in flight
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int ln = threadIdx.x & 63;
    asm volatile("" : "+v"(ln));
    issue(it + 3, (it + 3) & 3, ln);
    const int fr = ln & 15, g = ln >> 4;
    const int rbase = fr * 128 + ((g ^ fr) << 3);
    const int trow = 4 * g + (fr >> 2);
    const int tbase = trow * 128 + ((((ln & 3) >> 1) ^ trow) << 3) + (ln & 1) * 4;
    const bf16_t* Ks = ring + (it & 3) * DQ128_STAGE;
    const bf16_t* Vs = Ks + 4096;
    const int k0 = kstart + it * 32;
    const bool active = !CAUSAL || (k0 <= qw0 + 31 + off);
    if (active) {
      const bool need_mask = (qw0 + 32 > p.Sq) || (k0 + 32 > kvlen) || (CAUSAL && (k0 + 31 > qw0 + off)) || (p.window > 0);
      u32x2 dsk[2][2];                                 // [qt][kt] packed dS halves
      f32x4 st[2][2], dpt[2][2];                       // [kt][qt]
#define DQ_MF(KT)                                                                                             \
  _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) { st[KT][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; dpt[KT][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; } \
  _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                        \
    const bf16x8 ka = *(const bf16x8*)(Ks + (rbase ^ (ks * 32)) + (KT) * 2048);                               \
    const bf16x8 va = *(const bf16x8*)(Vs + (rbase ^ (ks * 32)) + (KT) * 2048);                               \
    _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                                        \
      st[KT][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[qt][ks], st[KT][qt], 0, 0, 0);              \
      dpt[KT][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[qt][ks], dpt[KT][qt], 0, 0, 0);           \
    }                                                                                                         \
  }
#define DQ_SM(KT, MASK)                                                                                           \
  _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
      float pv = fast_exp2(fmaf(st[KT][qt][r], c, -lse[qt]));                                                 \
      if (MASK) {                                                                                             \
        const int key = k0 + (KT) * 16 + 4 * g + r, qrow = qw0 + qt * 16 + fr;                                \
        const bool ok = qrow < p.Sq && key < kvlen && (!CAUSAL || key <= qrow + off) && (p.window <= 0 || key > qrow + off - p.window); \
        pv = ok ? pv : 0.f;                                                                                   \
      }                                                                                                       \
      dpt[KT][qt][r] = pv * (dpt[KT][qt][r] - dlt[qt]);                                                       \
    }                                                                                                         \
    dsk[qt][KT] = u32x2{pack_bf16x2(dpt[KT][qt][0], dpt[KT][qt][1]), pack_bf16x2(dpt[KT][qt][2], dpt[KT][qt][3])}; \
  }
      DQ_MF(0)
      __builtin_amdgcn_sched_barrier(0);
      // key half 1's MFMAs interleaved with key half 0's softmax arithmetic (different pipes): 1 MFMA : 5 VALU.  Two copies so that
      // each is ONE basic block (a branch on need_mask inside would fence the scheduler).
#define DQ_REGION(MASK)                                                                    \
  DQ_MF(1)                                                                                 \
  DQ_SM(0, MASK)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
    if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
    __builtin_amdgcn_sched_group_barrier(0x002, (MASK) ? 9 : 5, 0);                        \
  }                                                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                       \
  DQ_SM(1, MASK)                                                                           \
  __builtin_amdgcn_sched_barrier(0);
      if (need_mask) { DQ_REGION(true) } else { DQ_REGION(false) }
#undef DQ_REGION
#undef DQ_MF
#undef DQ_SM
      bf16x8 dsf[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        dsf[qt] = __builtin_bit_cast(bf16x8, u32x4{dsk[qt][0][0], dsk[qt][0][1], dsk[qt][1][0], dsk[qt][1][1]});
      const uint32_t ks_addr = attn_lds_addr(Ks);      // rows +16 = +4096 bytes
      s16x4 kl = tr_read_asm<0>(ks_addr + 2u * (uint32_t)tbase), kh = tr_read_asm<4096>(ks_addr + 2u * (uint32_t)tbase);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        s16x4 nl = kl, nh = kh;
        if (d + 1 < NDB) {                                // one fragment ahead
          const uint32_t na = ks_addr + 2u * (uint32_t)(tbase ^ ((d + 1) * 16));
          nl = tr_read_asm<0>(na);
          nh = tr_read_asm<4096>(na);
          ATTN_LGKM(2);
        } else {
          ATTN_LGKM(0);
        }
        bf16x8 ktf = tr_join(kl, kh);
        ATTN_PIN(ktf);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dq[qt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qt], dq[qt][d], 0, 0, 0);
        kl = nl;
        kh = nh;
        if (d & 1) __builtin_amdgcn_sched_barrier(0);
      }

    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (dummy) DMAs must not outlive the block's LDS
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = qw0 + qt * 16 + (lane & 15);
    if (qrow < p.Sq) {
      bf16_t* dqp = p.dq + (long)b * p.dq_bs + (long)qrow * p.dq_ts + (long)h * D;
      const long pp = ROPE ? (p.rope_pos ? (long)p.rope_pos[(long)b * p.Sq + qrow] : (long)(qrow + p.Skv - p.Sq)) * 64 : 0;
      store_row128<ROPE>(dqp, dq[qt], p.scale, lane >> 4, p.rope_cos + pp, p.rope_sin + pp);
    }
  }
}

// ================================================================================================
// forward for D = 128 (the decoder's shape).  Block = 4 waves x 32 query rows of one q head (two 16-row MFMA column tiles per wave, so
// every K / V fragment read from LDS feeds TWO MFMAs: the 8-wave x 16-row kernel above needs 256 B/clk of LDS reads per CU to keep the
// matrix pipe fed, twice what the LDS delivers); K / V stream through a 4-stage ring of 32-key tiles (global_load_lds, swizzled source
// chunks, counted vmcnt, one raw s_barrier per tile) like the backward kernels.  The loop is software-pipelined ACROSS tiles:
//   A(it): mask, running max, (rare) rescale                          -- small VALU block with the wave-uniform branch
//   B(it): S(it+1) = K(it+1) Q^T MFMAs  ||  exp2 / row sums / bf16 packing of tile it     -- ONE basic block, sched_group_barrier
//   C(it): O += V(it)^T P(it) MFMAs fed by transposing LDS reads
// so the matrix pipe works under the softmax arithmetic of the same wave, and the second wave of the SIMD (other block) fills the rest.
// ================================================================================================
constexpr int FWD128_STAGE = 2 * 32 * 128;            // bf16 units: K tile | V tile
constexpr int FWD128_LDS = 4 * FWD128_STAGE * 2;      // bytes

template <bool CAUSAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd128_kernel(AttnParams p) {
  constexpr int D = 128, NKS = 4, NDB = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  bf16_t* const ring = (bf16_t*)attn_smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (p.Sq + 127) >> 7;
  const int qb = nqb - 1 - VP_BZ(p);            // z is the slowest dispatch index: heavy (late) causal blocks first
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;     // whole GQA groups per XCD
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 128, qw0 = q0 + wave * 32;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;

  bf16x8 qf[2][NKS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrc = min(qw0 + qt * 16 + (lane & 15), p.Sq - 1);          // clamped (unconditional loads); rows >= Sq are never stored
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)qrc * p.q_ts + (long)h * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qt][ks] = *(const bf16x8*)(qp + ks * 32 + (lane >> 4) * 8);
  }
  f32x4 oacc[2][NDB];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int d = 0; d < NDB; ++d) oacc[qt][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};   // m is kept PRE-scaled: m = c * max(raw score)

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 128 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~31;
  const int nit = kend > kstart ? (kend - kstart + 31) / 32 : 0;
  // tiles this WAVE needs (the rest lie entirely above its causal diagonal): it <= last_w
  const int last_w = CAUSAL ? min(nit - 1, (qw0 + 31 + off - kstart) >> 5) : nit - 1;
  const char* kb = (const char*)(p.k + (long)b * p.k_bs + (long)hk * D);
  const char* vb = (const char*)(p.v + (long)b * p.v_bs + (long)hk * D);

  auto issue = [&](int t, int st, int ln) {            // lane-derived values are re-derived per call (nothing lane-derived stays live)
    const int k0 = kstart + min(t, nit - 1) * 32;
    const int drow = wave * 4 + (ln >> 4);
    const int dch = ((ln & 15) ^ (drow & 15)) * 8;
    bf16_t* sb = ring + st * FWD128_STAGE;
    const unsigned r0 = (unsigned)min(k0 + drow, p.Skv - 1), r1 = (unsigned)min(k0 + drow + 16, p.Skv - 1);
    const unsigned kts = (unsigned)p.k_ts, vts = (unsigned)p.v_ts;
    ATTN_GLDS(kb + (size_t)((r0 * kts + dch) * 2u), sb + wave * 512, 16);
    ATTN_GLDS(kb + (size_t)((r1 * kts + dch) * 2u), sb + (4 + wave) * 512, 16);
    ATTN_GLDS(vb + (size_t)((r0 * vts + dch) * 2u), sb + 4096 + wave * 512, 16);
    ATTN_GLDS(vb + (size_t)((r1 * vts + dch) * 2u), sb + 4096 + (4 + wave) * 512, 16);
  };
  f32x4 sc[2][2], sn[2][2];                            // [kt][qt] raw scores of the current / next tile
#define FWD_QK(DST, KS_PTR, RB)                                                                               \
  _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                          \
    _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) DST[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};                  \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                      \
      const bf16x8 ka = *(const bf16x8*)((KS_PTR) + ((RB) ^ (ks * 32)) + kt * 2048);                           \
      _Pragma("unroll") for (int qt = 0; qt < 2; ++qt)                                                        \
        DST[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[qt][ks], DST[kt][qt], 0, 0, 0);          \
    }                                                                                                         \
  }
  if (nit > 0) {
    issue(0, 0, lane); issue(1, 1, lane); issue(2, 2, lane);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    {
      const int fr = lane & 15, g = lane >> 4;
      const int rbase = fr * 128 + ((g ^ fr) << 3);
      FWD_QK(sc, ring, rbase)
    }
  }
  // One loop trip = two tiles with the roles of the two score buffers swapped (no register copies).  Inside a tile:
  //   A: mask (diagonal / ragged tiles only), LANE-LOCAL running max; the cross-lane row max (LDS-crossbar shuffles, ~100 cycles each with
  //      no MFMA of this wave in flight) is only formed when some lane sees a score above m + 2^8: with every score <= m + RESCALE_THR the
  //      exponentials stay <= 2^8, so m may lag the true running max (the final O / l ratio does not depend on m);
  //   B: MFMA j of S(next tile) followed by fma(element j) | exp2(element j-1) | add(element j-2) (no dependent pair adjacent), a
  //      scheduling fence after each slot; K fragments are read one k-step ahead; this copy of the softmax arithmetic is kept inside the
  //      block by routing the scale through an opaque asm (hipcc would otherwise hoist the code common to both branches);
  //   C: O += V^T P with the transposing V reads three fragments ahead of their MFMAs.
#define EL_REF(A, I) A[((I) >> 2) & 1][(I) >> 3][(I) & 3]
#define FWD_S1(A, I, CC) { EL_REF(A, I) = fmaf(EL_REF(A, I), CC, -m[(I) >> 3]); }
#define FWD_S2(A, I) { EL_REF(A, I) = fast_exp2(EL_REF(A, I)); }
#define FWD_S3(A, I) { rs[(I) >> 3][(I) & 1] += EL_REF(A, I); }
#define FWD_ITER(IT, SC, SN)                                                                                  \
  {                                                                                                           \
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0); \
    int ln = threadIdx.x & 63; \
    asm volatile("" : "+v"(ln)); \
    issue((IT) + 3, ((IT) + 3) & 3, ln); \
    const int fr = ln & 15, g = ln >> 4; \
    const int rbase = fr * 128 + ((g ^ fr) << 3); \
    const int trow = 4 * g + (fr >> 2); \
    const int tbase = trow * 128 + ((((ln & 3) >> 1) ^ trow) << 3) + (ln & 1) * 4; \
    const bf16_t* Vs = ring + ((IT) & 3) * FWD128_STAGE + 4096; \
    const bf16_t* Kn = ring + (((IT) + 1) & 3) * FWD128_STAGE; \
    const int k0 = kstart + (IT) * 32; \
    if ((IT) <= last_w) { \
      const bool need_mask = (qw0 + 32 > p.Sq) || (k0 + 32 > kvlen) || (CAUSAL && (k0 + 31 > qw0 + off)) || (p.window > 0); \
      if (need_mask) { \
        const int kl = kvlen - k0 - 4 * g; \
_Pragma("unroll") \
        for (int qt = 0; qt < 2; ++qt) { \
          const int dq_ = qw0 + qt * 16 + fr + off - k0 - 4 * g; \
          const int hi = CAUSAL ? min(kl, dq_ + 1) : kl; \
          const int lo = p.window > 0 ? dq_ - p.window : -1; \
_Pragma("unroll") \
          for (int kt = 0; kt < 2; ++kt) \
_Pragma("unroll") \
            for (int r = 0; r < 4; ++r) { \
              const int e = kt * 16 + r; \
              SC[kt][qt][r] = (e < hi && e > lo) ? SC[kt][qt][r] : -INFINITY; \
            } \
        } \
      } \
      float mloc[2]; \
_Pragma("unroll") \
      for (int qt = 0; qt < 2; ++qt) \
        mloc[qt] = c * fmaxf(fmaxf(fmaxf(SC[0][qt][0], SC[0][qt][1]), fmaxf(SC[0][qt][2], SC[0][qt][3])), \
                             fmaxf(fmaxf(SC[1][qt][0], SC[1][qt][1]), fmaxf(SC[1][qt][2], SC[1][qt][3]))); \
      if (!__all(mloc[0] <= m[0] + RESCALE_THR && mloc[1] <= m[1] + RESCALE_THR)) { \
_Pragma("unroll") \
        for (int qt = 0; qt < 2; ++qt) { \
          float v = mloc[qt]; \
          v = fmaxf(v, __shfl_xor(v, 16, 64)); \
          v = fmaxf(v, __shfl_xor(v, 32, 64)); \
          const float mnew = fmaxf(m[qt], v); \
          const float alpha = fast_exp2(m[qt] - mnew); \
          l[qt] *= alpha; \
_Pragma("unroll") \
          for (int d = 0; d < NDB; ++d) oacc[qt][d] *= alpha; \
          m[qt] = mnew; \
        } \
      } \
      const bool have_next = (IT) + 1 <= last_w; \
      u32x4 pk[2]; \
      float rs[2][2] = {{0.f, 0.f}, {0.f, 0.f}}; \
      __builtin_amdgcn_sched_barrier(0); \
      if (have_next) { \
        float c2 = c; \
        asm volatile("" : "+v"(c2)); \
        bf16x8 ka[2], kn[2]; \
        ka[0] = *(const bf16x8*)(Kn + rbase); \
        ka[1] = *(const bf16x8*)(Kn + rbase + 2048); \
_Pragma("unroll") \
        for (int ks = 0; ks < NKS; ++ks) { \
          if (ks + 1 < NKS) { \
            kn[0] = *(const bf16x8*)(Kn + (rbase ^ ((ks + 1) * 32))); \
            kn[1] = *(const bf16x8*)(Kn + (rbase ^ ((ks + 1) * 32)) + 2048); \
          } \
_Pragma("unroll") \
          for (int kt = 0; kt < 2; ++kt) \
_Pragma("unroll") \
            for (int qt = 0; qt < 2; ++qt) { \
              SN[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kt], qf[qt][ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : SN[kt][qt], 0, 0, 0); \
              FWD_S1(SC, ks * 4 + kt * 2 + qt, c2) \
              if (ks * 4 + kt * 2 + qt >= 1) FWD_S2(SC, (ks * 4 + kt * 2 + qt + 15) & 15) \
              if (ks * 4 + kt * 2 + qt >= 2) FWD_S3(SC, (ks * 4 + kt * 2 + qt + 14) & 15) \
              __builtin_amdgcn_sched_barrier(0); \
            } \
          ka[0] = kn[0]; \
          ka[1] = kn[1]; \
        } \
        FWD_S2(SC, 15) \
        FWD_S3(SC, 14) \
        FWD_S3(SC, 15) \
      } else { \
_Pragma("unroll") \
        for (int i = 0; i < 16; ++i) FWD_S1(SC, i, c) \
_Pragma("unroll") \
        for (int i = 0; i < 16; ++i) FWD_S2(SC, i) \
_Pragma("unroll") \
        for (int i = 0; i < 16; ++i) FWD_S3(SC, i) \
      } \
_Pragma("unroll") \
      for (int qt = 0; qt < 2; ++qt) { \
        l[qt] += rs[qt][0] + rs[qt][1]; \
        pk[qt] = u32x4{pack_bf16x2(SC[0][qt][0], SC[0][qt][1]), pack_bf16x2(SC[0][qt][2], SC[0][qt][3]), \
                       pack_bf16x2(SC[1][qt][0], SC[1][qt][1]), pack_bf16x2(SC[1][qt][2], SC[1][qt][3])}; \
      } \
      __builtin_amdgcn_sched_barrier(0); \
      const bf16x8 pf0 = __builtin_bit_cast(bf16x8, pk[0]), pf1 = __builtin_bit_cast(bf16x8, pk[1]); \
      const uint32_t vs_addr = attn_lds_addr(Vs); \
      constexpr int PF = 3; \
      s16x4 vlo[NDB], vhi[NDB]; \
_Pragma("unroll") \
      for (int d = 0; d < PF; ++d) { \
        const uint32_t na = vs_addr + 2u * (uint32_t)(tbase ^ (d * 16)); \
        vlo[d] = tr_read_asm<0>(na); \
        vhi[d] = tr_read_asm<4096>(na); \
      } \
_Pragma("unroll") \
      for (int d = 0; d < NDB; ++d) { \
        if (d + PF < NDB) { \
          const uint32_t na = vs_addr + 2u * (uint32_t)(tbase ^ ((d + PF) * 16)); \
          vlo[d + PF] = tr_read_asm<0>(na); \
          vhi[d + PF] = tr_read_asm<4096>(na); \
          ATTN_LGKM(2 * PF); \
        } else if (d + PF == NDB) { ATTN_LGKM(2 * (PF - 1)); } \
        else if (d + PF == NDB + 1) { ATTN_LGKM(PF >= 2 ? 2 * (PF - 2) : 0); } \
        else { ATTN_LGKM(0); } \
        bf16x8 vtf = tr_join(vlo[d], vhi[d]); \
        ATTN_PIN(vtf); \
        oacc[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf, pf0, oacc[0][d], 0, 0, 0); \
        oacc[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf, pf1, oacc[1][d], 0, 0, 0); \
        if (d & 1) __builtin_amdgcn_sched_barrier(0); \
      } \
    } \
  }
  for (int it = 0; it < nit; it += 2) {
    FWD_ITER(it, sc, sn)
    if (it + 1 < nit) FWD_ITER(it + 1, sn, sc)
  }
#undef FWD_ITER
#undef FWD_S1
#undef FWD_S2
#undef FWD_S3
#undef EL_REF
#undef FWD_QK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (dummy) DMAs must not outlive the block's LDS
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    l[qt] += __shfl_xor(l[qt], 16, 64);
    l[qt] += __shfl_xor(l[qt], 32, 64);
    const int qrow = qw0 + qt * 16 + (lane & 15);
    if (qrow < p.Sq) {
      const float inv = l[qt] > 0.f ? 1.f / l[qt] : 0.f;
      bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ts + (long)h * D;
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(oacc[qt][d][r] * inv);
        *(bf16x4*)(op + d * 16 + 4 * (lane >> 4)) = o;
      }
      if (p.lse && (lane >> 4) == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = (l[qt] > 0.f) ? m[qt] + log2f(l[qt]) : -1e30f;
    }
  }
}

// ================================================================================================
// forward for D = 128, round 3: 32x32x16 MFMAs with the swapped product S^T = K Q^T.
// Block = 8 waves x 32 query rows (256 rows of one q head, one block per CU, two waves per SIMD); K / V stream through a 4-stage ring of
// 64-key tiles (global_load_lds, swizzled source chunks, counted vmcnt, one raw s_barrier per tile) — every tile is fetched once per 256 rows.
//   * S^T[key][query]: the lane owns ONE query (lane & 31) and 32 of the tile's 64 scores of it (the other 32 sit in lane ^ 32): running max,
//     exp2, row sum are lane-local scalars; one cross-half exchange only when the max jumps by more than 2^8 (T13 defer-max).
//   * O^T[d][query] += V^T P^T: the P^T operand (16 keys x 32 queries, lane = query, 8 key slots) is a plain bf16 pack of 8 CONSECUTIVE
//     accumulator registers of S^T (register 8 t + s of key block kb <-> key 32 kb + 16 t + 8 (s >> 2) + 4 hh + (s & 3), hh = lane >> 5);
//     the V^T operand gathers the SAME keys per slot with two transposing reads (4 consecutive keys each), so no lane exchange is needed,
//     and O's columns stay with the lane that owns the query: the rescale is a per-lane scalar multiply.
//   * a 1 KB K or V fragment feeds a 32x32x16 MFMA = 16 K MACs (the 16-row kernel: 8 K) -> half the LDS bytes per flop.
// ================================================================================================
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t fwdm_u32x4s __attribute__((ext_vector_type(4)));
// ring: [K stage 0..3 (64 keys x 128 features, 16 KB each)] [V stage 0..3]: every K fragment address is one loop-invariant VGPR + a 16-bit
// immediate (stage, key block), every V fragment address likewise (the loop is unrolled by four, so the stage is a literal)
constexpr int FWDM_LDS = 8 * 64 * 128 * 2;            // bytes (128 KB)

template <bool CAUSAL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd128m_kernel(AttnParams p) {
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (p.Sq + 255) >> 8;
  const int qb = nqb - 1 - VP_BZ(p);                   // z is the slowest dispatch index: heavy (late) causal blocks first
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;     // whole GQA groups per XCD
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 256, qw0 = q0 + wave * 32;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;
  const int ql = lane & 31, hh = lane >> 5;
  const int qrow = qw0 + ql;

  bf16x8 qf[8];                                         // B operand of S^T: lane = query, 8 features at 16 ks + 8 hh
  {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)min(qrow, p.Sq - 1) * p.q_ts + (long)h * D;      // clamped; rows >= Sq are never stored
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16 + hh * 8);
  }
  f32x16 oacc[4];                                       // O^T: feature 32 db + 8 (i >> 2) + 4 hh + (i & 3) of this lane's query
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
  float m = -1e30f, l = 0.f;                            // m is kept PRE-scaled: m = c * max(raw score)

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 256 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~63;
  const int nit = kend > kstart ? (kend - kstart + 63) / 64 : 0;
  const int last_w = CAUSAL ? min(nit - 1, (qw0 + 31 + off - kstart) >> 6) : nit - 1;     // tiles this WAVE needs

  // ---- LDS-DMA: buffer_load ... lds with a per-(batch, kv head) descriptor (rows past Skv read as zeros: no clamps), ONE lane offset per
  // operand, the tile / piece in the scalar offset, the LDS destination in m0: no vector arithmetic per piece.  Piece 0 / 1 = K rows drow,
  // drow + 32 of the tile, 2 / 3 = V likewise (drow = 4 wave + lane / 16: a wave instruction fills 4 rows = 1 KB of LDS).
  const uint32_t ldsb = attn_lds_addr(attn_smem);
  auto make_rs = [&](const bf16_t* base, long ts) -> fwdm_u32x4s {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    fwdm_u32x4s r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)((((long)p.Skv - 1) * ts + D) * 2));
    r[3] = 0x00020000u;
    return r;
  };
  const fwdm_u32x4s rsK = make_rs(p.k + (long)b * p.k_bs + (long)hk * D, p.k_ts), rsV = make_rs(p.v + (long)b * p.v_bs + (long)hk * D, p.v_ts);
  uint32_t vK, vV;
  {
    const int drow = wave * 4 + (lane >> 4);
    // K rows: chunk ^ (row & 15): conflict-free ds_read_b128 (16 rows x one chunk per lane group).  V rows: chunk ^ 4 (row & 3): a transposing
    // read of this kernel touches 4 rows x 4 chunks x 2 halves per 32-lane group (two 16-lane groups share the rows and differ in the chunk);
    // with the K swizzle 16 (row, chunk) pairs fall on 4 slots (PMC: 6 conflict cycles per read), this way every lane has its own 8 bytes
    vK = (uint32_t)((drow * p.k_ts + (((lane & 15) ^ (drow & 15)) << 3)) * 2);
    vV = (uint32_t)((drow * p.v_ts + (((lane & 15) ^ ((drow & 3) << 2)) << 3)) * 2);
    asm volatile("" : "+v"(vK), "+v"(vV));
  }
  const uint32_t kts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.k_ts * 2)), vts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.v_ts * 2));
  const uint32_t m0w = __builtin_amdgcn_readfirstlane(ldsb + (uint32_t)wave * 1024u);
#define FWDM_DMA(T, ST, PIECE)                                                                                  \
  {                                                                                                             \
    const uint32_t row_ = (uint32_t)(kstart + (T) * 64 + (((PIECE) & 1) ? 32 : 0));                              \
    const uint32_t so_ = row_ * (((PIECE) & 2) ? vts2 : kts2);                                                   \
    const uint32_t m0_ = m0w + (uint32_t)((ST) * 16384 + (((PIECE) & 1) ? 8192 : 0) + (((PIECE) & 2) ? 65536 : 0)); \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_), "v"(((PIECE) & 2) ? vV : vK), \
                 "s"(((PIECE) & 2) ? rsV : rsK), "s"(so_) : "memory");                                          \
  }
#define FWDM_DMA4(T, ST) { FWDM_DMA(T, ST, 0) FWDM_DMA(T, ST, 1) FWDM_DMA(T, ST, 2) FWDM_DMA(T, ST, 3) }

  // ---- fragment addresses (bytes, loop-invariant).  K fragment (kb, ks): row 32 kb + (lane & 31), 16-byte chunk (2 ks + hh) ^ (row & 15)
  uint32_t ka[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    ka[ks] = ldsb + (uint32_t)(((lane & 31) * 128 + (((2 * ks + hh) ^ (lane & 15)) << 3)) * 2);
    asm volatile("" : "+v"(ka[ks]));
  }
  // V^T fragment (db, kt): two transposing reads; lane i of a 16-lane group supplies 4 features of key row 16 kt + 4 hh + (i >> 2) [+ 8]
  uint32_t va0[4];
  {
    const int fr_ = lane & 15, gq = (lane >> 4) & 1, trow = 4 * hh + (fr_ >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      va0[db] = ldsb + 65536u + (uint32_t)((trow * 128 + (((4 * db + 2 * gq + ((lane & 3) >> 1)) ^ ((trow & 3) << 2)) << 3) + (lane & 1) * 4) * 2);
      asm volatile("" : "+v"(va0[db]));
    }
  }
#define FWDM_KRD(DST, ST, N) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ka[(N) >> 1]), "n"((ST) * 16384 + ((N) & 1) * 8192))
#define FWDM_VRD(ST, N)                                                                                         \
  {                                                                                                             \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[N]) : "v"(va0[(N) & 3]), "n"((ST) * 16384 + ((N) >> 2) * 4096)); \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[N]) : "v"(va0[(N) & 3]), "n"((ST) * 16384 + ((N) >> 2) * 4096 + 2048)); \
  }

  f32x16 sa[2], sb2[2];                                 // S^T of the current / next tile: [key block]
  if (nit > 0) {
    FWDM_DMA4(0, 0) FWDM_DMA4(min(1, nit - 1), 1) FWDM_DMA4(min(2, nit - 1), 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    bf16x8 kf[16];
#define FWDM_P0(N) FWDM_KRD(kf[N], 0, N);
    FWDM_P0(0) FWDM_P0(1) FWDM_P0(2) FWDM_P0(3) FWDM_P0(4) FWDM_P0(5) FWDM_P0(6) FWDM_P0(7)
    FWDM_P0(8) FWDM_P0(9) FWDM_P0(10) FWDM_P0(11) FWDM_P0(12) FWDM_P0(13) FWDM_P0(14) FWDM_P0(15)
#undef FWDM_P0
    ATTN_LGKM(0);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      ATTN_PIN(kf[n]);
      if (n < 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sa[n & 1][i] = 0.f;
      }
      sa[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[n], qf[n >> 1], sa[n & 1], 0, 0, 0);
    }
  }
  // One tile.  STG = its ring stage (literal).  A: mask (diagonal / ragged tiles), lane-local max, rare rescale.  B: S^T of the NEXT tile
  // (16 MFMAs, two accumulator chains alternating, K fragments three steps ahead by asm reads with counted waits) with this tile's 32
  // exponentials in the MFMAs' shadow, 2 per MFMA — ALWAYS run, also behind the wave's last tile (a second code path for "no next tile"
  // made every score register a phi: 32 v_mov per tile).  C: O^T += V^T P^T (16 MFMAs, four chains alternating, transposing reads two steps
  // ahead), one LDS-DMA piece of tile it + 3 per four MFMAs.
#define FWDM_QK(SN, SC, NS, N)                                                                                  \
  {                                                                                                             \
    if ((N) + 3 < 16) { FWDM_KRD(kf[((N) + 3) & 15], NS, ((N) + 3) & 15); ATTN_LGKM(3); }                        \
    else if ((N) + 3 == 16) { ATTN_LGKM(2); }                                                                   \
    else if ((N) + 3 == 17) { ATTN_LGKM(1); }                                                                   \
    else { ATTN_LGKM(0); }                                                                                      \
    ATTN_PIN(kf[N]);                                                                                            \
    if ((N) < 2) { _Pragma("unroll") for (int i = 0; i < 16; ++i) SN[(N) & 1][i] = 0.f; }                       \
    SN[(N) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[N], qf[(N) >> 1], SN[(N) & 1], 0, 0, 0);            \
    {                                                                                                           \
      const float e0_ = fast_exp2(fmaf(SC[(N) >> 3][(2 * (N)) & 15], c2, -m));                                  \
      const float e1_ = fast_exp2(fmaf(SC[(N) >> 3][(2 * (N) + 1) & 15], c2, -m));                              \
      SC[(N) >> 3][(2 * (N)) & 15] = e0_;                                                                       \
      SC[(N) >> 3][(2 * (N) + 1) & 15] = e1_;                                                                   \
      rs0 += e0_;                                                                                               \
      rs1 += e1_;                                                                                               \
    }                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  }
#define FWDM_PV(ST, N)            /* step N: feature block N & 3, key group N >> 2 */                           \
  {                                                                                                             \
    if ((N) + 2 < 16) { FWDM_VRD(ST, ((N) + 2) & 15) ATTN_LGKM(4); }                                            \
    else if ((N) + 2 == 16) { ATTN_LGKM(2); }                                                                   \
    else { ATTN_LGKM(0); }                                                                                      \
    bf16x8 vtf = tr_join(vlo[N], vhi[N]);                                                                       \
    ATTN_PIN(vtf);                                                                                              \
    oacc[(N) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vtf, pf[(N) >> 2], oacc[(N) & 3], 0, 0, 0);          \
    if (((N) & 3) == 1) FWDM_DMA(t3_, ((ST) + 3) & 3, (N) >> 2)                                                 \
    if ((N) & 1) __builtin_amdgcn_sched_barrier(0);                                                             \
  }
#define FWDM_ITER(IT, STG, SC, SN)                                                                              \
  {                                                                                                             \
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    __builtin_amdgcn_s_barrier();                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    const int t3_ = min((IT) + 3, nit - 1);                                                                     \
    const int k0 = kstart + (IT) * 64;                                                                          \
    const bool need_mask = (k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw0 + off)) || (p.window > 0);            \
    if (need_mask) {                                                                                            \
      int ln = threadIdx.x & 63;                                                                                \
      asm volatile("" : "+v"(ln));                                                                              \
      const int dq_ = qw0 + (ln & 31) + off - k0 - 4 * (ln >> 5);                                               \
      const int kl = kvlen - k0 - 4 * (ln >> 5);                                                                \
      const int hi = CAUSAL ? min(kl, dq_ + 1) : kl;                                                            \
      const int lo = p.window > 0 ? dq_ - p.window : -1000000;                                                  \
      const float ninf_ = -INFINITY;                                                                            \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                        \
          const int e = kb * 32 + 8 * (i >> 2) + (i & 3);                                                       \
          const unsigned long long ok_ = __builtin_amdgcn_ballot_w64(e < hi && e > lo);                         \
          asm volatile("v_cndmask_b32 %0, %1, %0, %2" : "+v"(SC[kb][i]) : "v"(ninf_), "s"(ok_));                \
        }                                                                                                       \
    }                                                                                                           \
    float mx = vmax3(SC[0][0], SC[0][1], SC[0][2]);                                                             \
    _Pragma("unroll") for (int i = 3; i < 15; i += 2) mx = vmax3(mx, SC[0][i], SC[0][i + 1]);                   \
    mx = vmax3(mx, SC[0][15], SC[1][0]);                                                                        \
    _Pragma("unroll") for (int i = 1; i < 15; i += 2) mx = vmax3(mx, SC[1][i], SC[1][i + 1]);                   \
    mx = fmaxf(mx, SC[1][15]) * c;                                                                              \
    if (!__all(mx <= m + RESCALE_THR)) {                                                                        \
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                                                                   \
      const float mnew = fmaxf(m, mx);                                                                          \
      const float alpha = fast_exp2(m - mnew);                                                                  \
      l *= alpha;                                                                                               \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;                                     \
      m = mnew;                                                                                                 \
    }                                                                                                           \
    float rs0 = 0.f, rs1 = 0.f;                                                                                 \
    float c2 = c;                                                                                               \
    asm volatile("" : "+v"(c2));                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    {                                                                                                           \
      bf16x8 kf[16];                                                                                            \
      FWDM_KRD(kf[0], ((STG) + 1) & 3, 0); FWDM_KRD(kf[1], ((STG) + 1) & 3, 1); FWDM_KRD(kf[2], ((STG) + 1) & 3, 2); \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 0) FWDM_QK(SN, SC, ((STG) + 1) & 3, 1) FWDM_QK(SN, SC, ((STG) + 1) & 3, 2) FWDM_QK(SN, SC, ((STG) + 1) & 3, 3) \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 4) FWDM_QK(SN, SC, ((STG) + 1) & 3, 5) FWDM_QK(SN, SC, ((STG) + 1) & 3, 6) FWDM_QK(SN, SC, ((STG) + 1) & 3, 7) \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 8) FWDM_QK(SN, SC, ((STG) + 1) & 3, 9) FWDM_QK(SN, SC, ((STG) + 1) & 3, 10) FWDM_QK(SN, SC, ((STG) + 1) & 3, 11) \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 12) FWDM_QK(SN, SC, ((STG) + 1) & 3, 13) FWDM_QK(SN, SC, ((STG) + 1) & 3, 14) FWDM_QK(SN, SC, ((STG) + 1) & 3, 15) \
    }                                                                                                           \
    l += rs0 + rs1;                                                                                             \
    bf16x8 pf[4];                               /* P^T operands: (kb, t) = registers 8 t .. 8 t + 7 of key block kb */ \
    _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) {                                                          \
      const int kb = kt >> 1, t = kt & 1;                                                                       \
      pf[kt] = __builtin_bit_cast(bf16x8, u32x4{pack_bf16x2(SC[kb][8 * t + 0], SC[kb][8 * t + 1]), pack_bf16x2(SC[kb][8 * t + 2], SC[kb][8 * t + 3]), \
                                                pack_bf16x2(SC[kb][8 * t + 4], SC[kb][8 * t + 5]), pack_bf16x2(SC[kb][8 * t + 6], SC[kb][8 * t + 7])}); \
    }                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    {                                                                                                           \
      s16x4 vlo[16], vhi[16];                                                                                   \
      FWDM_VRD(STG, 0) FWDM_VRD(STG, 1)                                                                         \
      FWDM_PV(STG, 0) FWDM_PV(STG, 1) FWDM_PV(STG, 2) FWDM_PV(STG, 3) FWDM_PV(STG, 4) FWDM_PV(STG, 5) FWDM_PV(STG, 6) FWDM_PV(STG, 7) \
      FWDM_PV(STG, 8) FWDM_PV(STG, 9) FWDM_PV(STG, 10) FWDM_PV(STG, 11) FWDM_PV(STG, 12) FWDM_PV(STG, 13) FWDM_PV(STG, 14) FWDM_PV(STG, 15) \
    }                                                                                                           \
  }
  // the wave's own tiles (a conditional body inside ONE loop over all tiles cost 40 registers: every accumulator became a phi) ...
  if (last_w >= 0) {
    for (int it = 0;; it += 4) {
      FWDM_ITER(it, 0, sa, sb2)
      if (it + 1 > last_w) break;
      FWDM_ITER(it + 1, 1, sb2, sa)
      if (it + 2 > last_w) break;
      FWDM_ITER(it + 2, 2, sa, sb2)
      if (it + 3 > last_w) break;
      FWDM_ITER(it + 3, 3, sb2, sa)
      if (it + 4 > last_w) break;
    }
  }
  // ... then the tiles above its diagonal that the block's other waves still need: keep the barrier count and feed the ring
  for (int it = max(last_w, -1) + 1; it < nit; ++it) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int t3 = min(it + 3, nit - 1), st3 = (it + 3) & 3;
    const uint32_t so_k = (uint32_t)(kstart + t3 * 64) * kts2, so_v = (uint32_t)(kstart + t3 * 64) * vts2;
    const uint32_t m0_ = m0w + (uint32_t)st3 * 16384u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_), "v"(vK), "s"(rsK), "s"(so_k) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_ + 8192u), "v"(vK), "s"(rsK), "s"(so_k + 32u * kts2) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_ + 65536u), "v"(vV), "s"(rsV), "s"(so_v) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_ + 65536u + 8192u), "v"(vV), "s"(rsV), "s"(so_v + 32u * vts2) : "memory");
  }
#undef FWDM_ITER
#undef FWDM_PV
#undef FWDM_QK
#undef FWDM_VRD
#undef FWDM_KRD
#undef FWDM_DMA4
#undef FWDM_DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (dummy) DMAs must not outlive the block's LDS
  l += __shfl_xor(l, 32, 64);
  if (qrow < p.Sq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ts + (long)h * D;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(oacc[db][4 * j + r] * inv);
        *(bf16x4*)(op + db * 32 + 8 * j + 4 * hh) = o;
      }
    if (p.lse && hh == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = (l > 0.f) ? m + log2f(l) : -1e30f;
  }
}

// ================================================================================================
// forward for D = 128, one wave per SIMD (VP_ATTN_FWDQ=1): the attn_fwd128m structure with 4 waves x 64 query rows — a K / V fragment feeds TWO
// MFMAs, no partner wave competes for the matrix pipe — and the register file split by who touches what: O^T (128 registers) and the Q
// fragments (64) live in AGPRs and are only ever MFMA operands (inline-asm MFMAs with "a" constraints: left to itself hipcc parked VALU-touched
// values in AGPRs and moved 477 registers per tile back and forth); the two S^T buffers, P^T and everything the softmax touches stay in VGPRs.
// The rare rescale of O^T goes through v_accvgpr_read / write pairs.
// ================================================================================================
template <bool CAUSAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_fwd128q_kernel(AttnParams p) {
  constexpr int NQ = 2;
  constexpr int NW = 8 / NQ;
  constexpr int NPK = 2 * NQ;
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (p.Sq + 255) >> 8;
  const int qb = nqb - 1 - VP_BZ(p);                   // z is the slowest dispatch index: heavy (late) causal blocks first
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;     // whole GQA groups per XCD
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qb * 256, qw0 = q0 + wave * 32 * NQ;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq;
  const float c = p.scale * LOG2E;
  const int ql = lane & 31, hh = lane >> 5;

  bf16x8 qf[NQ][8];
#pragma unroll
  for (int qb_ = 0; qb_ < NQ; ++qb_) {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)min(qw0 + qb_ * 32 + ql, p.Sq - 1) * p.q_ts + (long)h * D;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[qb_][ks] = *(const bf16x8*)(qp + ks * 16 + hh * 8);
  }
#pragma unroll
  for (int qb_ = 0; qb_ < NQ; ++qb_)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[qb_][ks]));       // from here on an AGPR tuple (only "a"-constrained uses follow)
  f32x16 oacc[NQ][4];
#pragma unroll
  for (int qb_ = 0; qb_ < NQ; ++qb_)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[qb_][d][i] = 0.f;
#pragma unroll
  for (int qb_ = 0; qb_ < NQ; ++qb_)
#pragma unroll
    for (int d = 0; d < 4; ++d) asm volatile("" : "+a"(oacc[qb_][d]));
  float m[NQ], l[NQ];
#pragma unroll
  for (int qb_ = 0; qb_ < NQ; ++qb_) { m[qb_] = -1e30f; l[qb_] = 0.f; }

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 256 + off);
  int kstart = 0;
  if (p.window > 0) kstart = max(0, (q0 + off - p.window + 1)) & ~63;
  const int nit = kend > kstart ? (kend - kstart + 63) / 64 : 0;
  const int last_w = CAUSAL ? min(nit - 1, (qw0 + 32 * NQ - 1 + off - kstart) >> 6) : nit - 1;     // tiles this WAVE needs

  // ---- LDS-DMA: buffer_load ... lds with a per-(batch, kv head) descriptor (rows past Skv read as zeros: no clamps), ONE lane offset per
  // operand, the tile / piece in the scalar offset, the LDS destination in m0: no vector arithmetic per piece.  Piece 0 / 1 = K rows drow,
  // drow + 32 of the tile, 2 / 3 = V likewise (drow = 4 wave + lane / 16: a wave instruction fills 4 rows = 1 KB of LDS).
  const uint32_t ldsb = attn_lds_addr(attn_smem);
  auto make_rs = [&](const bf16_t* base, long ts) -> fwdm_u32x4s {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    fwdm_u32x4s r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)((((long)p.Skv - 1) * ts + D) * 2));
    r[3] = 0x00020000u;
    return r;
  };
  const fwdm_u32x4s rsK = make_rs(p.k + (long)b * p.k_bs + (long)hk * D, p.k_ts), rsV = make_rs(p.v + (long)b * p.v_bs + (long)hk * D, p.v_ts);
  uint32_t vK, vV;
  {
    const int drow = wave * 4 + (lane >> 4);
    // K rows: chunk ^ (row & 15): conflict-free ds_read_b128 (16 rows x one chunk per lane group).  V rows: chunk ^ 4 (row & 3): a transposing
    // read of this kernel touches 4 rows x 4 chunks x 2 halves per 32-lane group (two 16-lane groups share the rows and differ in the chunk);
    // with the K swizzle 16 (row, chunk) pairs fall on 4 slots (PMC: 6 conflict cycles per read), this way every lane has its own 8 bytes
    vK = (uint32_t)((drow * p.k_ts + (((lane & 15) ^ (drow & 15)) << 3)) * 2);
    vV = (uint32_t)((drow * p.v_ts + (((lane & 15) ^ ((drow & 3) << 2)) << 3)) * 2);
    asm volatile("" : "+v"(vK), "+v"(vV));
  }
  const uint32_t kts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.k_ts * 2)), vts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.v_ts * 2));
  const uint32_t m0w = __builtin_amdgcn_readfirstlane(ldsb + (uint32_t)wave * 1024u);
#define FWDQ_QK0(S_, KF_, QA_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(S_) : "v"(KF_), "a"(QA_))
#define FWDQ_QK(S_, KF_, QA_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S_) : "v"(KF_), "a"(QA_))
#define FWDQ_PV(O_, VF_, PF_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(O_) : "v"(VF_), "v"(PF_))
#define FWDM_DMA(T, ST, PIECE)                                                                                  \
  {                                                                                                             \
    const uint32_t row_ = (uint32_t)(kstart + (T) * 64 + ((PIECE) % NPK) * NW * 4);                              \
    const uint32_t so_ = row_ * (((PIECE) >= NPK) ? vts2 : kts2);                                                \
    const uint32_t m0_ = m0w + (uint32_t)((ST) * 16384 + ((PIECE) % NPK) * NW * 1024 + (((PIECE) >= NPK) ? 65536 : 0)); \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_), "v"(((PIECE) >= NPK) ? vV : vK), \
                 "s"(((PIECE) >= NPK) ? rsV : rsK), "s"(so_) : "memory");                                       \
  }
#define FWDM_DMA_ALL(T, ST) { _Pragma("unroll") for (int pc_ = 0; pc_ < 2 * NPK; ++pc_) FWDM_DMA(T, ST, pc_) }

  // ---- fragment addresses (bytes, loop-invariant).  K fragment (kb, ks): row 32 kb + (lane & 31), 16-byte chunk (2 ks + hh) ^ (row & 15)
  uint32_t ka[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    ka[ks] = ldsb + (uint32_t)(((lane & 31) * 128 + (((2 * ks + hh) ^ (lane & 15)) << 3)) * 2);
    asm volatile("" : "+v"(ka[ks]));
  }
  // V^T fragment (db, kt): two transposing reads; lane i of a 16-lane group supplies 4 features of key row 16 kt + 4 hh + (i >> 2) [+ 8]
  uint32_t va0[4];
  {
    const int fr_ = lane & 15, gq = (lane >> 4) & 1, trow = 4 * hh + (fr_ >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      va0[db] = ldsb + 65536u + (uint32_t)((trow * 128 + (((4 * db + 2 * gq + ((lane & 3) >> 1)) ^ ((trow & 3) << 2)) << 3) + (lane & 1) * 4) * 2);
      asm volatile("" : "+v"(va0[db]));
    }
  }
#define FWDM_KRD(DST, ST, N) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ka[(N) >> 1]), "n"((ST) * 16384 + ((N) & 1) * 8192))
#define FWDM_VRD(ST, N)                                                                                         \
  {                                                                                                             \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[N]) : "v"(va0[(N) & 3]), "n"((ST) * 16384 + ((N) >> 2) * 4096)); \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[N]) : "v"(va0[(N) & 3]), "n"((ST) * 16384 + ((N) >> 2) * 4096 + 2048)); \
  }

  f32x16 sa[NQ][2], sb2[NQ][2];
  if (nit > 0) {
    FWDM_DMA_ALL(0, 0) FWDM_DMA_ALL(min(1, nit - 1), 1) FWDM_DMA_ALL(min(2, nit - 1), 2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NPK) : "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 kf[16];
#define FWDM_P0(N) FWDM_KRD(kf[N], 0, N);
    FWDM_P0(0) FWDM_P0(1) FWDM_P0(2) FWDM_P0(3) FWDM_P0(4) FWDM_P0(5) FWDM_P0(6) FWDM_P0(7)
    FWDM_P0(8) FWDM_P0(9) FWDM_P0(10) FWDM_P0(11) FWDM_P0(12) FWDM_P0(13) FWDM_P0(14) FWDM_P0(15)
#undef FWDM_P0
    ATTN_LGKM(0);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      ATTN_PIN(kf[n]);
#pragma unroll
      for (int qb_ = 0; qb_ < NQ; ++qb_) {
        if (n < 2) FWDQ_QK0(sa[qb_][n & 1], kf[n], qf[qb_][n >> 1]);
        else FWDQ_QK(sa[qb_][n & 1], kf[n], qf[qb_][n >> 1]);
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMAs: the compiler inserts no MFMA -> VALU wait states itself
  }
  // One tile.  STG = its ring stage (literal).  A: mask (diagonal / ragged tiles), lane-local max, rare rescale.  B: S^T of the NEXT tile
  // (16 MFMAs, two accumulator chains alternating, K fragments three steps ahead by asm reads with counted waits) with this tile's 32
  // exponentials in the MFMAs' shadow, 2 per MFMA — ALWAYS run, also behind the wave's last tile (a second code path for "no next tile"
  // made every score register a phi: 32 v_mov per tile).  C: O^T += V^T P^T (16 MFMAs, four chains alternating, transposing reads two steps
  // ahead), one LDS-DMA piece of tile it + 3 per four MFMAs.
#define FWDM_QK(SN, SC, NS, N)                                                                                  \
  {                                                                                                             \
    if ((N) + 3 < 16) { FWDM_KRD(kf[((N) + 3) & 15], NS, ((N) + 3) & 15); ATTN_LGKM(3); }                        \
    else if ((N) + 3 == 16) { ATTN_LGKM(2); }                                                                   \
    else if ((N) + 3 == 17) { ATTN_LGKM(1); }                                                                   \
    else { ATTN_LGKM(0); }                                                                                      \
    ATTN_PIN(kf[N]);                                                                                            \
    _Pragma("unroll") for (int qb_ = 0; qb_ < NQ; ++qb_) {                                                      \
      if ((N) < 2) FWDQ_QK0(SN[qb_][(N) & 1], kf[N], qf[qb_][(N) >> 1]);                                         \
      else FWDQ_QK(SN[qb_][(N) & 1], kf[N], qf[qb_][(N) >> 1]);                                                 \
      const float e0_ = fast_exp2(fmaf(SC[qb_][(N) >> 3][(2 * (N)) & 15], c2, -m[qb_]));                        \
      const float e1_ = fast_exp2(fmaf(SC[qb_][(N) >> 3][(2 * (N) + 1) & 15], c2, -m[qb_]));                    \
      SC[qb_][(N) >> 3][(2 * (N)) & 15] = e0_;                                                                  \
      SC[qb_][(N) >> 3][(2 * (N) + 1) & 15] = e1_;                                                              \
      rs0[qb_] += e0_;                                                                                          \
      rs1[qb_] += e1_;                                                                                          \
      if (NQ == 2) __builtin_amdgcn_sched_barrier(0);                                                           \
    }                                                                                                           \
    if (NQ == 1) __builtin_amdgcn_sched_barrier(0);                                                             \
  }
#define FWDM_PV(ST, N)            /* step N: feature block N & 3, key group N >> 2 */                           \
  {                                                                                                             \
    if ((N) + 2 < 16) { FWDM_VRD(ST, ((N) + 2) & 15) ATTN_LGKM(4); }                                            \
    else if ((N) + 2 == 16) { ATTN_LGKM(2); }                                                                   \
    else { ATTN_LGKM(0); }                                                                                      \
    bf16x8 vtf = tr_join(vlo[N], vhi[N]);                                                                       \
    ATTN_PIN(vtf);                                                                                              \
    _Pragma("unroll") for (int qb_ = 0; qb_ < NQ; ++qb_)                                                        \
      FWDQ_PV(oacc[qb_][(N) & 3], vtf, pf[qb_][(N) >> 2]);                                                      \
    if (((N) % (4 / NQ)) == 1 % (4 / NQ)) FWDM_DMA(t3_, ((ST) + 3) & 3, (N) / (4 / NQ))                          \
    if (((N) & 1) || NQ == 2) __builtin_amdgcn_sched_barrier(0);                                                \
  }
#define FWDM_ITER(IT, STG, SC, SN)                                                                              \
  {                                                                                                             \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPK) : "memory");                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    __builtin_amdgcn_s_barrier();                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    const int t3_ = min((IT) + 3, nit - 1);                                                                     \
    const int k0 = kstart + (IT) * 64;                                                                          \
    const bool need_mask = (k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw0 + off)) || (p.window > 0);            \
    float mx[NQ];                                                                                               \
    _Pragma("unroll") for (int qb_ = 0; qb_ < NQ; ++qb_) {                                                      \
      if (need_mask) {                                                                                          \
        int ln = threadIdx.x & 63;                                                                              \
        asm volatile("" : "+v"(ln));                                                                            \
        const int dq_ = qw0 + qb_ * 32 + (ln & 31) + off - k0 - 4 * (ln >> 5);                                  \
        const int kl = kvlen - k0 - 4 * (ln >> 5);                                                              \
        const int hi = CAUSAL ? min(kl, dq_ + 1) : kl;                                                          \
        const int lo = p.window > 0 ? dq_ - p.window : -1000000;                                                \
        const float ninf_ = -INFINITY;                                                                          \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                        \
          _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                      \
            const int e = kb * 32 + 8 * (i >> 2) + (i & 3);                                                     \
            const unsigned long long ok_ = __builtin_amdgcn_ballot_w64(e < hi && e > lo);                       \
            asm volatile("v_cndmask_b32 %0, %1, %0, %2" : "+v"(SC[qb_][kb][i]) : "v"(ninf_), "s"(ok_));         \
          }                                                                                                     \
      }                                                                                                         \
      float mx_ = vmax3(SC[qb_][0][0], SC[qb_][0][1], SC[qb_][0][2]);                                           \
      _Pragma("unroll") for (int i = 3; i < 15; i += 2) mx_ = vmax3(mx_, SC[qb_][0][i], SC[qb_][0][i + 1]);     \
      mx_ = vmax3(mx_, SC[qb_][0][15], SC[qb_][1][0]);                                                          \
      _Pragma("unroll") for (int i = 1; i < 15; i += 2) mx_ = vmax3(mx_, SC[qb_][1][i], SC[qb_][1][i + 1]);     \
      mx[qb_] = fmaxf(mx_, SC[qb_][1][15]) * c;                                                                 \
    }                                                                                                           \
    bool calm_ = mx[0] <= m[0] + RESCALE_THR;                                                                   \
    if (NQ == 2) calm_ = calm_ && (mx[NQ - 1] <= m[NQ - 1] + RESCALE_THR);                                      \
    if (!__all(calm_)) {                                                                                        \
      _Pragma("unroll") for (int qb_ = 0; qb_ < NQ; ++qb_) {                                                    \
        const float mxx = fmaxf(mx[qb_], __shfl_xor(mx[qb_], 32, 64));                                          \
        const float mnew = fmaxf(m[qb_], mxx);                                                                  \
        const float alpha = fast_exp2(m[qb_] - mnew);                                                           \
        l[qb_] *= alpha;                                                                                        \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                     \
        _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                           \
          _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                      \
            float t_;                                                                                           \
            asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_mul_f32 %0, %2, %0\n\tv_accvgpr_write_b32 %1, %0" : "=&v"(t_), "+a"(oacc[qb_][d][i]) : "v"(alpha)); \
          }                                                                                                     \
        m[qb_] = mnew;                                                                                          \
      }                                                                                                         \
    }                                                                                                           \
    float rs0[NQ], rs1[NQ];                                                                                     \
    _Pragma("unroll") for (int qb_ = 0; qb_ < NQ; ++qb_) { rs0[qb_] = 0.f; rs1[qb_] = 0.f; }                    \
    float c2 = c;                                                                                               \
    asm volatile("" : "+v"(c2));                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    {                                                                                                           \
      bf16x8 kf[16];                                                                                            \
      FWDM_KRD(kf[0], ((STG) + 1) & 3, 0); FWDM_KRD(kf[1], ((STG) + 1) & 3, 1); FWDM_KRD(kf[2], ((STG) + 1) & 3, 2); \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 0) FWDM_QK(SN, SC, ((STG) + 1) & 3, 1) FWDM_QK(SN, SC, ((STG) + 1) & 3, 2) FWDM_QK(SN, SC, ((STG) + 1) & 3, 3) \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 4) FWDM_QK(SN, SC, ((STG) + 1) & 3, 5) FWDM_QK(SN, SC, ((STG) + 1) & 3, 6) FWDM_QK(SN, SC, ((STG) + 1) & 3, 7) \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 8) FWDM_QK(SN, SC, ((STG) + 1) & 3, 9) FWDM_QK(SN, SC, ((STG) + 1) & 3, 10) FWDM_QK(SN, SC, ((STG) + 1) & 3, 11) \
      FWDM_QK(SN, SC, ((STG) + 1) & 3, 12) FWDM_QK(SN, SC, ((STG) + 1) & 3, 13) FWDM_QK(SN, SC, ((STG) + 1) & 3, 14) FWDM_QK(SN, SC, ((STG) + 1) & 3, 15) \
    }                                                                                                           \
    bf16x8 pf[NQ][4];                                                                                           \
    _Pragma("unroll") for (int qb_ = 0; qb_ < NQ; ++qb_) {                                                      \
      l[qb_] += rs0[qb_] + rs1[qb_];                                                                            \
      _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) {                                                        \
        const int kb = kt >> 1, t = kt & 1;                                                                     \
        pf[qb_][kt] = __builtin_bit_cast(bf16x8, u32x4{pack_bf16x2(SC[qb_][kb][8 * t + 0], SC[qb_][kb][8 * t + 1]), pack_bf16x2(SC[qb_][kb][8 * t + 2], SC[qb_][kb][8 * t + 3]), \
                                                       pack_bf16x2(SC[qb_][kb][8 * t + 4], SC[qb_][kb][8 * t + 5]), pack_bf16x2(SC[qb_][kb][8 * t + 6], SC[qb_][kb][8 * t + 7])}); \
      }                                                                                                         \
    }                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    {                                                                                                           \
      s16x4 vlo[16], vhi[16];                                                                                   \
      FWDM_VRD(STG, 0) FWDM_VRD(STG, 1)                                                                         \
      FWDM_PV(STG, 0) FWDM_PV(STG, 1) FWDM_PV(STG, 2) FWDM_PV(STG, 3) FWDM_PV(STG, 4) FWDM_PV(STG, 5) FWDM_PV(STG, 6) FWDM_PV(STG, 7) \
      FWDM_PV(STG, 8) FWDM_PV(STG, 9) FWDM_PV(STG, 10) FWDM_PV(STG, 11) FWDM_PV(STG, 12) FWDM_PV(STG, 13) FWDM_PV(STG, 14) FWDM_PV(STG, 15) \
    }                                                                                                           \
  }
  // the wave's own tiles (a conditional body inside ONE loop over all tiles cost 40 registers: every accumulator became a phi) ...
  if (last_w >= 0) {
    for (int it = 0;; it += 4) {
      FWDM_ITER(it, 0, sa, sb2)
      if (it + 1 > last_w) break;
      FWDM_ITER(it + 1, 1, sb2, sa)
      if (it + 2 > last_w) break;
      FWDM_ITER(it + 2, 2, sa, sb2)
      if (it + 3 > last_w) break;
      FWDM_ITER(it + 3, 3, sb2, sa)
      if (it + 4 > last_w) break;
    }
  }
  // ... then the tiles above its diagonal that the block's other waves still need: keep the barrier count and feed the ring
  for (int it = max(last_w, -1) + 1; it < nit; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPK) : "memory");
    __builtin_amdgcn_s_barrier();
    const int t3 = min(it + 3, nit - 1), st3 = (it + 3) & 3;
#pragma unroll
    for (int pc = 0; pc < 2 * NPK; ++pc) {
      const uint32_t row_ = (uint32_t)(kstart + t3 * 64 + (pc % NPK) * NW * 4);
      const uint32_t so_ = row_ * ((pc >= NPK) ? vts2 : kts2);
      const uint32_t m0_ = m0w + (uint32_t)st3 * 16384u + (uint32_t)((pc % NPK) * NW * 1024 + ((pc >= NPK) ? 65536 : 0));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0_), "v"((pc >= NPK) ? vV : vK), "s"((pc >= NPK) ? rsV : rsK), "s"(so_) : "memory");
    }
  }
#undef FWDM_ITER
#undef FWDM_PV
#undef FWDM_QK
#undef FWDM_VRD
#undef FWDM_KRD
#undef FWDM_DMA_ALL
#undef FWDQ_QK0
#undef FWDQ_QK
#undef FWDQ_PV
#undef FWDM_DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (dummy) DMAs must not outlive the block's LDS
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int qb_ = 0; qb_ < NQ; ++qb_) {
    const int qrow = qw0 + qb_ * 32 + ql;
    const float lt = l[qb_] + __shfl_xor(l[qb_], 32, 64);
    if (qrow < p.Sq) {
      const float inv = lt > 0.f ? 1.f / lt : 0.f;
      bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ts + (long)h * D;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t_;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t_) : "a"(oacc[qb_][db][4 * j + r]));
            o[r] = (short)f2bf(t_ * inv);
          }
          *(bf16x4*)(op + db * 32 + 8 * j + 4 * hh) = o;
        }
      if (p.lse && hh == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = (lt > 0.f) ? m[qb_] + log2f(lt) : -1e30f;
    }
  }
}

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

static bool vp_fwd128_enabled() {                      // VP_ATTN_FWD128=1 selects the DMA-ring forward (measured equal to the default: DESIGN.md 4)
  static int v = -1;
  if (v < 0) { const char* e = getenv("VP_ATTN_FWD128"); v = e ? atoi(e) : 0; }
  return v != 0;
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
  static int fwdq = -1;
  if (fwdq < 0) { const char* e = getenv("VP_ATTN_FWDQ"); fwdq = e ? atoi(e) : 0; }
  if (D == 128 && !p.bias_h && !p.bias_b && fwdq) {                     // one wave per SIMD, O^T / Q in AGPRs (see attn_fwd128q_kernel)
    static bool attrq = false;
    if (!attrq) {
      (void)hipFuncSetAttribute((const void*)attn_fwd128q_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FWDM_LDS);
      (void)hipFuncSetAttribute((const void*)attn_fwd128q_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FWDM_LDS);
      attrq = true;
    }
    const int nb = (p.Sq + 255) / 256;
    const dim3 gq = p.order ? dim3(p.Hq, nb, p.B) : dim3(p.Hq, p.B, nb);
    if (causal) hipLaunchKernelGGL((attn_fwd128q_kernel<true>), gq, dim3(256), FWDM_LDS, s, p);
    else hipLaunchKernelGGL((attn_fwd128q_kernel<false>), gq, dim3(256), FWDM_LDS, s, p);
    return vp_check_launch("vp_attn_fwd");
  }
  if (D == 128 && !p.bias_h && !p.bias_b && vp_fwdm_enabled()) {        // round 3: 32x32x16 swapped-product kernel, 256-row blocks
    static bool attrm = false;
    if (!attrm) {
      (void)hipFuncSetAttribute((const void*)attn_fwd128m_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FWDM_LDS);
      (void)hipFuncSetAttribute((const void*)attn_fwd128m_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FWDM_LDS);
      attrm = true;
    }
    const int nb = (p.Sq + 255) / 256;
    const dim3 gm = p.order ? dim3(p.Hq, nb, p.B) : dim3(p.Hq, p.B, nb);
    if (causal) hipLaunchKernelGGL((attn_fwd128m_kernel<true>), gm, dim3(512), FWDM_LDS, s, p);
    else hipLaunchKernelGGL((attn_fwd128m_kernel<false>), gm, dim3(512), FWDM_LDS, s, p);
    return vp_check_launch("vp_attn_fwd");
  }
  if (D == 128 && !p.bias_h && !p.bias_b && vp_fwd128_enabled()) {      // DMA-ring kernel (32 query rows per wave)
    static bool attr128 = false;
    if (!attr128) {
      (void)hipFuncSetAttribute((const void*)attn_fwd128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_fwd128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD128_LDS);
      attr128 = true;
    }
    if (causal) hipLaunchKernelGGL((attn_fwd128_kernel<true>), grid, dim3(256), FWD128_LDS, s, p);
    else hipLaunchKernelGGL((attn_fwd128_kernel<false>), grid, dim3(256), FWD128_LDS, s, p);
    return vp_check_launch("vp_attn_fwd");
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
template <int D>
static int launch_bwd(const AttnParams& p_in, int causal, hipStream_t s) {
  AttnParams p = p_in;
  p.order = (p.B <= 65535 && (p.Skv + 127) / 128 <= 65535 && (p.Sq + 127) / 128 <= 65535) ? vp_attn_order() : 0;
  const long rows = (long)p.B * p.Hq * p.Sq;
  if (D != 128) hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)min(8192L, (rows + 15) / 16)), dim3(256), 0, s, p);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds_bytes<D>());
    attr = true;
  }
  const dim3 g1 = p.order ? dim3(p.Hkv, (p.Skv + 127) / 128, p.B) : dim3(p.Hkv, p.B, (p.Skv + 127) / 128);
  const dim3 g2 = p.order ? dim3(p.Hq, (p.Sq + 127) / 128, p.B) : dim3(p.Hq, p.B, (p.Sq + 127) / 128);
  if (D == 128) {
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv128_kernel<true, DKDV_KT>, hipFuncAttributeMaxDynamicSharedMemorySize, DKDV128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv128_kernel<false, DKDV_KT>, hipFuncAttributeMaxDynamicSharedMemorySize, DKDV128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
      attr2 = true;
    }
    if (p.rope_cos) {                                   // fused RoPE backward: its own instantiations (training: causal only)
      if (!causal) { vp_set_error("vp_attn_bwd_rope: causal attention only"); return VP_ERR_UNSUPPORTED_SHAPE; }
      static bool attr3 = false;
      if (!attr3) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv128_kernel<true, DKDV_KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKDV128_LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ128_LDS);
        attr3 = true;
      }
      hipLaunchKernelGGL((attn_bwd_dq128_kernel<true, true>), g2, dim3(256), DQ128_LDS, s, p);          // dQ first: it also writes delta
      hipLaunchKernelGGL((attn_bwd_dkdv128_kernel<true, DKDV_KT, true>), g1, dim3(64 * (8 / DKDV_KT)), DKDV128_LDS, s, p);
    } else if (causal) {
      hipLaunchKernelGGL((attn_bwd_dq128_kernel<true>), g2, dim3(256), DQ128_LDS, s, p);
      hipLaunchKernelGGL((attn_bwd_dkdv128_kernel<true, DKDV_KT>), g1, dim3(64 * (8 / DKDV_KT)), DKDV128_LDS, s, p);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq128_kernel<false>), g2, dim3(256), DQ128_LDS, s, p);
      hipLaunchKernelGGL((attn_bwd_dkdv128_kernel<false, DKDV_KT>), g1, dim3(64 * (8 / DKDV_KT)), DKDV128_LDS, s, p);
    }
  } else if (causal) {
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, true>), g1, dim3(512), 0, s, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, true>), g2, dim3(512), kv_lds_bytes<D>(), s, p);
  } else {
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, false>), g1, dim3(512), 0, s, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false>), g2, dim3(512), kv_lds_bytes<D>(), s, p);
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
int vp_debug_attn_stamps(long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(vp_attn_dbg), sizeof(long) * 16) == hipSuccess ? 0 : 1; }

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
// to q, k before SDPA -- modeling_llama.py; its autograd rotates dq / dk back).  D = 128, causal only.  rope_cos / rope_sin: fp32
// [positions, 64] as for vp_rope; rope_pos: int32 [B, S] position ids or NULL (position = row index).  Same numbers as vp_attn_bwd
// followed by vp_rope(inverse = 1) on dq and dk.
int vp_attn_bwd_rope(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k, long k_bs,
                     long k_ts, const void* v, long v_bs, long v_ts, const void* o, long o_bs, long o_ts, const float* lse,
                     const void* dout, long do_bs, long do_ts, void* dq, long dq_bs, long dq_ts, void* dk, long dk_bs, long dk_ts,
                     void* dv, long dv_bs, long dv_ts, float* delta, const int* kv_len, int causal, int window, float scale,
                     const float* rope_cos, const float* rope_sin, const int* rope_pos, hipStream_t s) {
  int e = check_attn("vp_attn_bwd_rope", B, Hq, Hkv, Sq, Skv, D);
  if (e) return e;
  VP_REQUIRE(lse && delta && dout && dq && dk && dv && rope_cos && rope_sin, VP_ERR_BAD_ARG, "vp_attn_bwd_rope: null pointer");
  VP_REQUIRE(D == 128 && causal, VP_ERR_UNSUPPORTED_SHAPE, "vp_attn_bwd_rope: head_dim 128, causal only (got %d, causal %d)", D, causal);
  VP_REQUIRE(((((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15) == 0, VP_ERR_BAD_ARG, "vp_attn_bwd_rope: tables must be 16-byte aligned");
  AttnParams p{};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = (float*)lse;
  p.dout = (const bf16_t*)dout; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.delta = delta;
  p.q_bs = q_bs; p.q_ts = q_ts; p.k_bs = k_bs; p.k_ts = k_ts; p.v_bs = v_bs; p.v_ts = v_ts; p.o_bs = o_bs; p.o_ts = o_ts;
  p.do_bs = do_bs; p.do_ts = do_ts; p.dq_bs = dq_bs; p.dq_ts = dq_ts; p.dk_bs = dk_bs; p.dk_ts = dk_ts; p.dv_bs = dv_bs; p.dv_ts = dv_ts;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv; p.window = window; p.kv_len = kv_len; p.scale = scale;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_pos = rope_pos;
  return launch_bwd<128>(p, causal, s);
}

}  // extern "C"
