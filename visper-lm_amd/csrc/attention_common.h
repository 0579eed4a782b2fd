// Shared declarations of the attention kernels: parameter block, small device helpers, the register-staged tile loader.
#pragma once
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
  // 1: the dQ kernel writes the per-row statistics as two planes (-lse / c at delta[0 .. rows), -delta at delta[rows .. 2 rows)) for the round-5 dK/dV
  // kernel (attention_bwd64.h) instead of round 4's interleaved (lse, delta) pairs
  int stat_planes;
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

