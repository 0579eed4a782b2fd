// The D = 128 class of attention kernels (the Llama decoder's shape): LDS-DMA ring kernels for the backward pass (dK/dV, dQ) and the
// 32x32x16 swapped-product forward.  Included by attention.hip (one translation unit: the launchers live there).
#pragma once
#include "attention_common.h"

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
// LDS-DMA through a buffer descriptor (`buffer_load ... lds`): wave-uniform descriptor (base of the tensor slice, num_records in BYTES: everything
// past it reads as zeros, also when the scalar offset carries the position past the end — checked on gfx950 with tools/probes/buf_range_probe.hip),
// ONE loop-invariant 32-bit lane offset per operand, the tile's row / head offset in the scalar offset, the LDS destination in m0: no 64-bit
// address arithmetic and no row clamps per piece.
typedef uint32_t fwdm_u32x4s __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ fwdm_u32x4s attn_make_rs(const void* base, long bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  fwdm_u32x4s r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  r[3] = 0x00020000u;
  return r;
}
#define ATTN_DMA16(M0, VOFF, RS, SOFF)                                                                          \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(M0), "v"(VOFF), "s"(RS), "s"(SOFF) : "memory")
#define ATTN_DMA4(M0, VOFF, RS, SOFF)                                                                           \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(M0), "v"(VOFF), "s"(RS), "s"(SOFF) : "memory")
#define ATTN_PIN(F) asm volatile("" : "+v"(F))

// store one row's D = 16 NDB features (this lane: blocks d = 0..NDB-1, features d*16 + 4g + r) as bf16, optionally rotated back by RoPE^T:
// x1 = feature f < D/2, x2 = feature f + D/2; out1 = bf(bf(x1 c) + bf(x2 s)), out2 = bf(bf(x2 c) + bf(-x1 s))  (vp_rope with inverse = 1).
// cs / sn: this row's D/2 cosines / sines (NDB = 8: the Llama shape; NDB = 6: Phi-3's D = 96, pairs 48 features apart)
template <bool ROPE, int NDB = 8>
static __device__ __forceinline__ void store_row128(bf16_t* dst, const f32x4 (&acc)[NDB], float scale, int g, const float* cs, const float* sn) {
  static_assert(NDB % 2 == 0, "the fused RoPE^T store pairs feature f with f + D/2");
  if constexpr (!ROPE) {
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
      bf16x4 a;
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = (short)f2bf(acc[d][r] * scale);
      *(bf16x4*)(dst + d * 16 + 4 * g) = a;
    }
  } else {
    constexpr int HB = NDB / 2;
#pragma unroll
    for (int d = 0; d < HB; ++d) {
      const f32x4 c4 = *(const f32x4*)(cs + d * 16 + 4 * g), s4 = *(const f32x4*)(sn + d * 16 + 4 * g);
      bf16x4 a, b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = bfround(acc[d][r] * scale), x2 = bfround(acc[d + HB][r] * scale);
        const float ss = -s4[r];
        a[r] = (short)f2bf(bfround(x1 * c4[r]) + bfround(-x2 * ss));
        b[r] = (short)f2bf(bfround(x2 * c4[r]) + bfround(x1 * ss));
      }
      *(bf16x4*)(dst + d * 16 + 4 * g) = a;
      *(bf16x4*)(dst + HB * 16 + d * 16 + 4 * g) = b;
    }
  }
}
#ifndef DKDV_KT
#define DKDV_KT 2
#endif
constexpr int DKDV128_STAGE = 2 * 32 * 128 + 128;     // bf16 units: Q tile | dO tile | 32 (lse, delta) fp32 pairs
constexpr int DKDV128_LDS = 4 * DKDV128_STAGE * 2;    // bytes

// D = 128 or 96 (Phi-3): the LDS tiles keep 128-feature rows and the same swizzle; with D = 96 the contraction runs over 3 instead of 4 k-steps
// and 6 instead of 8 output feature blocks, so the 4 LDS chunks per row that hold no feature of this head are never read (their DMA lanes
// re-fetch chunk 0: always inside the tensor).
template <bool CAUSAL, int KT, bool ROPE = false, int D = 128>      // KT = 16-key column tiles per wave: 2 -> 4 waves x 32 keys, 1 -> 8 waves x 16 keys (128 keys per block)
__global__ __launch_bounds__(64 * (8 / KT)) __attribute__((amdgpu_waves_per_eu(KT == 2 ? 2 : 4, KT == 2 ? 2 : 4)))
void attn_bwd_dkdv128_kernel(AttnParams p) {
  static_assert(D == 128 || D == 96, "D");
  constexpr int NKS = D / 32, NDB = D / 16, NW = 8 / KT, NI = 8 / NW;      // NI = DMA instructions per wave per 32-row tile
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
  // Pieces go through buffer descriptors (one per batch element, covering every head and row of q / dO / the (lse, delta) pairs: rows past Sq read
  // as zeros) with the tile's row and head in the SCALAR offset; the lane offsets are re-derived from the lane id per call behind an opaque asm
  // (32-bit, ~6 VALU): the register budget is full (dk/dv 128 + K/V fragments 64), and hipcc would otherwise spill loop invariants to scratch and
  // reload them with s_waitcnt vmcnt(0) -- which drains the DMA queue every iteration.  The flattened (head, q tile) index is walked by two
  // scalar counters instead of a division per tile.
  const fwdm_u32x4s rsQ = attn_make_rs(p.q + (long)b * p.q_bs, (((long)p.Sq - 1) * p.q_ts + (long)p.Hq * D) * 2);
  const fwdm_u32x4s rsG = attn_make_rs(p.dout + (long)b * p.do_bs, (((long)p.Sq - 1) * p.do_ts + (long)p.Hq * D) * 2);
  const fwdm_u32x4s rsP = attn_make_rs(pairs + (long)b * p.Hq * p.Sq * 2, (long)p.Hq * p.Sq * 2 * 4);
  const uint32_t ldsb = attn_lds_addr(attn_smem);
  const uint32_t qts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.q_ts * 2)), gts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.do_ts * 2));
  int ih = 0, iq = 0;                                  // (head within the group, q tile) of the NEXT tile to issue
  auto issue = [&](int st, int ln) {                   // (past the end: a harmless re-fetch of the last tile keeps the vmcnt bookkeeping uniform)
    const uint32_t h = (uint32_t)(hk * rep + ih), q0 = (uint32_t)(qstart + iq * 32);
    const uint32_t soQ = q0 * qts2 + h * (uint32_t)(D * 2), soG = q0 * gts2 + h * (uint32_t)(D * 2);
    const uint32_t m0s = ldsb + (uint32_t)(st * DKDV128_STAGE * 2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int blk = i * NW + wave;                   // 4-row group of the tile this instruction fills
      const int drow = blk * 4 + (ln >> 4);
      int dch = ((ln & 15) ^ (drow & 15)) * 8;
      if (D < 128 && dch >= D) dch = 0;                  // (a chunk no fragment read ever touches: keep its source inside the head)
      const uint32_t vq = (uint32_t)drow * qts2 + (uint32_t)dch * 2u, vg = (uint32_t)drow * gts2 + (uint32_t)dch * 2u;
      ATTN_DMA16(m0s + (uint32_t)(blk * 1024), vq, rsQ, soQ);
      ATTN_DMA16(m0s + (uint32_t)(8192 + blk * 1024), vg, rsG, soG);
    }
    if (wave == 0) ATTN_DMA4(m0s + 16384u, (uint32_t)ln * 4u, rsP, (h * (uint32_t)p.Sq + q0) * 8u);
    if (ih * ntq + iq + 1 < nit) { if (++iq == ntq) { iq = 0; ++ih; } }
  };
  if (nit > 0) { issue(0, lane); issue(1, lane); issue(2, lane); }
  int cq = 0;                                          // q tile (within its head) of the tile being consumed
  for (int it = 0; it < nit; ++it) {
    // tile `it` landed (this wave's part): at most the two younger stages may still be in flight
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (2 * NI + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NI) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                      // every wave's part landed; stage (it-1)&3 is no longer being read
    __builtin_amdgcn_sched_barrier(0);
    int ln = threadIdx.x & 63;
    asm volatile("" : "+v"(ln));                       // opaque: see the note above `issue`
    issue((it + 3) & 3, ln);
    const int fr = ln & 15, g = ln >> 4;
    // LDS fragment addressing: the swizzle is an XOR on the chunk bits, so one base per read kind + compile-time XOR masks
    const int rbase = fr * 128 + ((g ^ fr) << 3);                         // row-wise: (row fr, chunk g) ^ (ks*4 chunks), + qt*16 rows
    const int trow = 4 * g + (fr >> 2);
    const int tbase = trow * 128 + ((((ln & 3) >> 1) ^ trow) << 3) + (ln & 1) * 4;
    const bf16_t* Qs = ring + (it & 3) * DKDV128_STAGE;
    const bf16_t* dOs = Qs + 4096;
    const float* ld = (const float*)(Qs + 8192);
    const int q0 = qstart + cq * 32;
    if (++cq == ntq) cq = 0;
    // wave-uniform skip: every query of this tile is below this wave's first key (causal) -> all p = 0
    // ... or (sliding window) every key of the wave is at or below the tile's FIRST query's lower bound
    const bool active = (!CAUSAL || (q0 + 31 + off >= kw0)) && !(p.window > 0 && kw0 + 16 * KT - 1 <= q0 + off - p.window);
    if (active) {
      // s[qt][kt][r]: query = q0 + 16qt + 4g + r, key = kw0 + 16kt + fr.  Per tile (two 16-query halves):
      //   A  S / dP of half 0                                        (8 KT MFMAs)
      //   B  S / dP of half 1  ||  softmax + dS arithmetic of half 0 (8 KT MFMAs over ~45 VALU, placed by sched_group_barrier)
      //   C  softmax + dS arithmetic of half 1                       (VALU)
      //   D  dV += dO^T P, dK += Q^T dS per feature block            (16 KT MFMAs, transposing reads one block ahead)
      // ONE softmax code path: the mask is a v_cndmask per element against wave masks in SGPRs, all-ones in the (common) unmasked tile; only
      // their computation sits behind the wave-uniform branch.  What does not work here (round 4, measured): a branch on need_mask per element
      // fences the scheduler (a chain of 16 four-instruction blocks behind every MFMA burst: no overlap in B); two copies of the tile per mask
      // mode turn all 128 dk / dv accumulators into phis (215 spilled VGPRs); two copies of the softmax arithmetic alone cost 25 VGPRs of phis
      // = spilled K / V fragments, reloaded behind vmcnt(0), which drains the DMA ring; mask bounds kept live across the tile: 4 VGPRs too many.
      // (sliding window: only a tile that touches the window's lower edge needs the per-element test: key > query + off - window holds for every
      // pair as soon as the tile's first key is above its last query's bound)
      const bool need_mask = (q0 + 32 > p.Sq) || (kw0 + 16 * KT > kvlen) || (CAUSAL && (kw0 + 16 * KT - 1 > q0 + off)) ||
                             (p.window > 0 && kw0 <= q0 + 31 + off - p.window);
      const uint32_t qs_addr = attn_lds_addr(Qs);      // dO tile = +8192 bytes, rows +16 = +4096 bytes
      {
        f32x4 s[2][KT], dp[2][KT];
        u32x2 pk[KT][2], dsk[KT][2];
        auto mf = [&](int qt) __attribute__((always_inline)) {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) { s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 qa = *(const bf16x8*)(Qs + (rbase ^ (ks * 32)) + qt * 2048);
            const bf16x8 da = *(const bf16x8*)(dOs + (rbase ^ (ks * 32)) + qt * 2048);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], s[qt][kt], 0, 0, 0);
              dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[qt][kt], 0, 0, 0);
            }
          }
        };
        auto sm_p = [&](int qt) __attribute__((always_inline)) {
          unsigned long long okm[KT][4];
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) okm[kt][r] = ~0ull;
          if (need_mask) {
            // bounds of this lane, derived HERE from an opaque lane id (kept live across the tile they cost 4 VGPRs -> K/V fragment spills):
            // key = kw0 + 16kt + fr valid iff 16kt < m_kl; query = q0 + 16qt + 4g + r valid iff 16qt + r < m_ql;
            // causal: key <= query + off iff 16kt - 16qt - r <= m_dq; window: key > query + off - window iff 16kt - 16qt - r > m_lo
            int l2 = threadIdx.x & 63;
            asm volatile("" : "+v"(l2));
            const int fr2 = l2 & 15, g2 = l2 >> 4;
            const int m_kl = kvlen - kw0 - fr2, m_ql = p.Sq - q0 - 4 * g2, m_dq = q0 + 4 * g2 + off - kw0 - fr2;
            const int m_lo = p.window > 0 ? m_dq - p.window : -(1 << 30);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int e = kt * 16 - qt * 16 - r;           // key - query, up to the lane's base
                bool ok = (kt * 16 < m_kl) & (qt * 16 + r < m_ql) & (e > m_lo);
                if (CAUSAL) ok = ok & (e <= m_dq);
                okm[kt][r] = __builtin_amdgcn_ballot_w64(ok);
              }
          }
          const f32x4 l0 = *(const f32x4*)(ld + (qt * 16 + 4 * g) * 2), l1 = *(const f32x4*)(ld + (qt * 16 + 4 * g) * 2 + 4);
          const float lse_r[4] = {l0[0], l0[2], l1[0], l1[2]};
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pv = fast_exp2(fmaf(s[qt][kt][r], c, -lse_r[r]));
              asm("v_cndmask_b32 %0, 0, %0, %1" : "+v"(pv) : "s"(okm[kt][r]));
              s[qt][kt][r] = pv;
            }
            pk[kt][qt] = u32x2{pack_bf16x2(s[qt][kt][0], s[qt][kt][1]), pack_bf16x2(s[qt][kt][2], s[qt][kt][3])};
          }
        };
        auto sm_ds = [&](int qt) __attribute__((always_inline)) {
          const f32x4 l0 = *(const f32x4*)(ld + (qt * 16 + 4 * g) * 2), l1 = *(const f32x4*)(ld + (qt * 16 + 4 * g) * 2 + 4);
          const float del_r[4] = {l0[1], l0[3], l1[1], l1[3]};
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[qt][kt][r] = s[qt][kt][r] * (dp[qt][kt][r] - del_r[r]);
            dsk[kt][qt] = u32x2{pack_bf16x2(dp[qt][kt][0], dp[qt][kt][1]), pack_bf16x2(dp[qt][kt][2], dp[qt][kt][3])};
          }
        };
        // ---- A: S / dP of query half 0;  B: S / dP of half 1 under the softmax + dS arithmetic of half 0 (one MFMA, one LDS read, a few VALU)
        __builtin_amdgcn_s_setprio(1);
        mf(0);
        __builtin_amdgcn_sched_barrier(0);
        mf(1);
        sm_p(0);
        sm_ds(0);
#pragma unroll
        for (int i = 0; i < 2 * KT * NKS; ++i) {
          if (i < 2 * NKS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- C, then D: transposing reads one feature block ahead of the MFMAs that consume them (counted lgkmcnt: the block's own 4 reads are
        // the oldest)
        sm_p(1);
        sm_ds(1);
        bf16x8 pf[KT], dsf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          pf[kt] = __builtin_bit_cast(bf16x8, u32x4{pk[kt][0][0], pk[kt][0][1], pk[kt][1][0], pk[kt][1][1]});
          dsf[kt] = __builtin_bit_cast(bf16x8, u32x4{dsk[kt][0][0], dsk[kt][0][1], dsk[kt][1][0], dsk[kt][1][1]});
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);                  // MFMA bursts at raised priority: the co-resident block's VALU/LDS work yields (-3 %)
        s16x4 al, ah, ql, qh;
        {
          const uint32_t ta_ = qs_addr + 2u * (uint32_t)tbase;
          al = tr_read_asm<8192>(ta_); ah = tr_read_asm<12288>(ta_);
          ql = tr_read_asm<0>(ta_); qh = tr_read_asm<4096>(ta_);
        }
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          s16x4 nal = al, nah = ah, nql = ql, nqh = qh;
          if (d + 1 < NDB) {
            const uint32_t tn_ = qs_addr + 2u * (uint32_t)(tbase ^ ((d + 1) * 16));
            nal = tr_read_asm<8192>(tn_); nah = tr_read_asm<12288>(tn_);
            nql = tr_read_asm<0>(tn_); nqh = tr_read_asm<4096>(tn_);
            ATTN_LGKM(6);
          } else {
            ATTN_LGKM(2);
          }
          bf16x8 ta = tr_join(al, ah);
          ATTN_PIN(ta);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) dv[kt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta, pf[kt], dv[kt][d], 0, 0, 0);
          if (d + 1 < NDB) { ATTN_LGKM(4); } else { ATTN_LGKM(0); }
          bf16x8 tq = tr_join(ql, qh);
          ATTN_PIN(tq);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) dk[kt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq, dsf[kt], dk[kt][d], 0, 0, 0);
          al = nal; ah = nah; ql = nql; qh = nqh;
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing (dummy) DMAs must not outlive the block's LDS
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + kt * 16 + fr;
    if (key < p.Skv) {
      bf16_t* dkp = p.dk + (long)b * p.dk_bs + (long)key * p.dk_ts + (long)hk * D;
      bf16_t* dvp = p.dv + (long)b * p.dv_bs + (long)key * p.dv_ts + (long)hk * D;
      const long pp = ROPE ? (p.rope_pos ? (long)p.rope_pos[(long)b * p.Skv + key] : (long)key) * (D / 2) : 0;
      store_row128<ROPE, NDB>(dkp, dk[kt], p.scale, g, p.rope_cos + pp, p.rope_sin + pp);
      store_row128<false, NDB>(dvp, dv[kt], 1.f, g, nullptr, nullptr);
    }
  }
}

// ================================================================================================
// backward: dQ for D = 128.  Same structure as the DMA-fed dK/dV kernel: block = 4 waves x 32 queries of one q head,
// K / V stream through a 4-stage LDS ring in 32-key tiles (global_load_lds, swizzled source chunks, counted vmcnt).
// ================================================================================================
constexpr int DQ128_STAGE = 2 * 32 * 128;             // bf16 units: K tile | V tile
constexpr int DQ128_LDS = 4 * DQ128_STAGE * 2;        // bytes

template <bool CAUSAL, bool ROPE = false, int D = 128>      // D = 128 or 96 (see attn_bwd_dkdv128_kernel)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq128_kernel(AttnParams p) {
  static_assert(D == 128 || D == 96, "D");
  constexpr int NKS = D / 32, NDB = D / 16;
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
      if (p.stat_planes) {                               // round-5 dK/dV kernel: planes (-lse / c, -delta)
        p.delta[sidx] = -lse[qt] / c;
        p.delta[(long)p.B * p.Hq * p.Sq + sidx] = -part;
      } else {
        p.delta[sidx] = part;
        *(float2*)(p.delta + (long)p.B * p.Hq * p.Sq + 2 * sidx) = float2{lse[qt], part};
      }
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
  // K / V pieces through buffer descriptors of this (batch, kv head) (rows past Skv read as zeros), the tile's row offset in the scalar offset
  const fwdm_u32x4s rsK = attn_make_rs(p.k + (long)b * p.k_bs + (long)hk * D, (((long)p.Skv - 1) * p.k_ts + D) * 2);
  const fwdm_u32x4s rsV = attn_make_rs(p.v + (long)b * p.v_bs + (long)hk * D, (((long)p.Skv - 1) * p.v_ts + D) * 2);
  const uint32_t ldsb = attn_lds_addr(attn_smem);
  const uint32_t kts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.k_ts * 2)), vts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.v_ts * 2));

  auto issue = [&](int t, int st, int ln) {            // lane-derived values are re-derived per call (see the dK/dV kernel)
    const uint32_t k0 = (uint32_t)(kstart + min(t, nit - 1) * 32);
    const int drow = wave * 4 + (ln >> 4);
    int dch = ((ln & 15) ^ (drow & 15)) * 8;
    if (D < 128 && dch >= D) dch = 0;                    // (never read: see attn_bwd_dkdv128_kernel)
    const uint32_t vk = (uint32_t)drow * kts2 + (uint32_t)dch * 2u, vv = (uint32_t)drow * vts2 + (uint32_t)dch * 2u;
    const uint32_t m0s = ldsb + (uint32_t)(st * DQ128_STAGE * 2 + wave * 1024);
    ATTN_DMA16(m0s, vk, rsK, k0 * kts2);
    ATTN_DMA16(m0s + 4096u, vk, rsK, (k0 + 16u) * kts2);
    ATTN_DMA16(m0s + 8192u, vv, rsV, k0 * vts2);
    ATTN_DMA16(m0s + 8192u + 4096u, vv, rsV, (k0 + 16u) * vts2);
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
    const bool active = (!CAUSAL || (k0 <= qw0 + 31 + off)) && !(p.window > 0 && k0 + 31 <= qw0 + off - p.window);
    if (active) {
      const bool need_mask = (qw0 + 32 > p.Sq) || (k0 + 32 > kvlen) || (CAUSAL && (k0 + 31 > qw0 + off)) ||
                             (p.window > 0 && k0 <= qw0 + 31 + off - p.window);        // (window: see the dK/dV kernel)
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
      const long pp = ROPE ? (p.rope_pos ? (long)p.rope_pos[(long)b * p.Sq + qrow] : (long)(qrow + p.Skv - p.Sq)) * (D / 2) : 0;
      store_row128<ROPE, NDB>(dqp, dq[qt], p.scale, lane >> 4, p.rope_cos + pp, p.rope_sin + pp);
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
// ring: [K stage 0..3 (64 keys x 128 features, 16 KB each)] [V stage 0..3]: every K fragment address is one loop-invariant VGPR + a 16-bit
// immediate (stage, key block), every V fragment address likewise (the loop is unrolled by four, so the stage is a literal)
constexpr int FWDM_LDS = 8 * 64 * 128 * 2;            // bytes (128 KB)

// D = 128 or 96 (Phi-3): LDS rows stay 128 features wide with the same swizzle; with D = 96 a tile takes 12 instead of 16 MFMAs in each of the
// two products (6 k-steps x 2 key blocks; 3 feature blocks x 4 key groups), the 4 chunks per row that belong to no feature of this head are
// never read (their DMA lanes fetch the neighbouring head's bytes, or zeros past the end: the buffer descriptor's range check).
template <bool CAUSAL, int D = 128>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd128m_kernel(AttnParams p) {
  static_assert(D == 128 || D == 96, "D");
  constexpr int NQS = D / 16;                            // k-steps of S^T = K Q^T
  constexpr int NQK = 2 * NQS;                           // its MFMA steps per tile: step n = (k-step n >> 1, key block n & 1)
  constexpr int NOB = D / 32;                            // 32-feature blocks of O^T
  constexpr int NPV = 4 * NOB;                           // MFMA steps of O^T += V^T P^T: step n = (feature block n % NOB, key group n / NOB)
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

  bf16x8 qf[NQS];                                       // B operand of S^T: lane = query, 8 features at 16 ks + 8 hh
  {
    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)min(qrow, p.Sq - 1) * p.q_ts + (long)h * D;      // clamped; rows >= Sq are never stored
#pragma unroll
    for (int ks = 0; ks < NQS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16 + hh * 8);
  }
  f32x16 oacc[NOB];                                     // O^T: feature 32 db + 8 (i >> 2) + 4 hh + (i & 3) of this lane's query
#pragma unroll
  for (int d = 0; d < NOB; ++d)
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
  uint32_t ka[NQS];
#pragma unroll
  for (int ks = 0; ks < NQS; ++ks) {
    ka[ks] = ldsb + (uint32_t)(((lane & 31) * 128 + (((2 * ks + hh) ^ (lane & 15)) << 3)) * 2);
    asm volatile("" : "+v"(ka[ks]));
  }
  // V^T fragment (db, kt): two transposing reads; lane i of a 16-lane group supplies 4 features of key row 16 kt + 4 hh + (i >> 2) [+ 8]
  uint32_t va0[NOB];
  {
    const int fr_ = lane & 15, gq = (lane >> 4) & 1, trow = 4 * hh + (fr_ >> 2);
#pragma unroll
    for (int db = 0; db < NOB; ++db) {
      va0[db] = ldsb + 65536u + (uint32_t)((trow * 128 + (((4 * db + 2 * gq + ((lane & 3) >> 1)) ^ ((trow & 3) << 2)) << 3) + (lane & 1) * 4) * 2);
      asm volatile("" : "+v"(va0[db]));
    }
  }
#define FWDM_KRD(DST, ST, N) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ka[(N) >> 1]), "n"((ST) * 16384 + ((N) & 1) * 8192))
#define FWDM_VRD(ST, N)                                                                                         \
  {                                                                                                             \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[N]) : "v"(va0[(N) % NOB]), "n"((ST) * 16384 + ((N) / NOB) * 4096)); \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[N]) : "v"(va0[(N) % NOB]), "n"((ST) * 16384 + ((N) / NOB) * 4096 + 2048)); \
  }

  f32x16 sa[2], sb2[2];                                 // S^T of the current / next tile: [key block]
  if (nit > 0) {
    FWDM_DMA4(0, 0) FWDM_DMA4(min(1, nit - 1), 1) FWDM_DMA4(min(2, nit - 1), 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    bf16x8 kf[16];
#define FWDM_P0(N) if constexpr ((N) < NQK) FWDM_KRD(kf[N], 0, N);
    FWDM_P0(0) FWDM_P0(1) FWDM_P0(2) FWDM_P0(3) FWDM_P0(4) FWDM_P0(5) FWDM_P0(6) FWDM_P0(7)
    FWDM_P0(8) FWDM_P0(9) FWDM_P0(10) FWDM_P0(11) FWDM_P0(12) FWDM_P0(13) FWDM_P0(14) FWDM_P0(15)
#undef FWDM_P0
    ATTN_LGKM(0);
#pragma unroll
    for (int n = 0; n < NQK; ++n) {
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
  if constexpr ((N) < NQK) {                                                                                    \
    if constexpr ((N) + 3 < NQK) { FWDM_KRD(kf[((N) + 3) & 15], NS, ((N) + 3) & 15); ATTN_LGKM(3); }             \
    else if constexpr ((N) + 3 == NQK) { ATTN_LGKM(2); }                                                        \
    else if constexpr ((N) + 3 == NQK + 1) { ATTN_LGKM(1); }                                                    \
    else { ATTN_LGKM(0); }                                                                                      \
    ATTN_PIN(kf[N]);                                                                                            \
    if ((N) < 2) { _Pragma("unroll") for (int i = 0; i < 16; ++i) SN[(N) & 1][i] = 0.f; }                       \
    SN[(N) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[N], qf[(N) >> 1], SN[(N) & 1], 0, 0, 0);            \
    /* this tile's 32 exponentials spread over the NQK steps: 2 per step (D = 128), 2-3 per step (D = 96); even ones sum into rs0 */ \
    _Pragma("unroll") for (int e_ = (32 * (N)) / NQK; e_ < (32 * ((N) + 1)) / NQK; ++e_) {                      \
      const float ev_ = fast_exp2(fmaf(SC[e_ >> 4][e_ & 15], c2, -m));                                          \
      SC[e_ >> 4][e_ & 15] = ev_;                                                                               \
      if (e_ & 1) rs1 += ev_; else rs0 += ev_;                                                                  \
    }                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  }
#define FWDM_PV(ST, N)            /* step N: feature block N % NOB, key group N / NOB */                        \
  if constexpr ((N) < NPV) {                                                                                    \
    if constexpr ((N) + 2 < NPV) { FWDM_VRD(ST, ((N) + 2) & 15) ATTN_LGKM(4); }                                  \
    else if constexpr ((N) + 2 == NPV) { ATTN_LGKM(2); }                                                        \
    else { ATTN_LGKM(0); }                                                                                      \
    bf16x8 vtf = tr_join(vlo[N], vhi[N]);                                                                       \
    ATTN_PIN(vtf);                                                                                              \
    oacc[(N) % NOB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vtf, pf[(N) / NOB], oacc[(N) % NOB], 0, 0, 0);    \
    if constexpr (((N) % NOB) == 1) FWDM_DMA(t3_, ((ST) + 3) & 3, (N) / NOB)                                    \
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
    const bool need_mask = (k0 + 64 > kvlen) || (CAUSAL && (k0 + 63 > qw0 + off)) || (p.window > 0 && k0 <= qw0 + 31 + off - p.window); \
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
      _Pragma("unroll") for (int d = 0; d < NOB; ++d)                                                           \
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
  // O leaves as WHOLE rows (round 4): the lane that owns a query holds its D features 4 at a time, and a row-per-lane store (16 dwordx2 per
  // lane, 32 rows x 16 B per instruction) moves ~7 B/clk per CU while all 8 waves of the CU's one block wait for it (tools/probes/
  // store_pattern_probe.hip: partial-line stores 15.8, whole-line stores 52.9 B/clk).  The finished ring is the staging buffer: every wave
  // writes its 32 x D tile into a private 8 KB slice (8-byte unit u of row r at position u ^ r: conflict-free for the 16-lane write groups
  // and for the row-wise reads), reads it back 4 rows per instruction and stores 16 bytes per lane: 4 whole rows per buffer instruction.
  __builtin_amdgcn_s_barrier();                        // every wave is done reading K / V: the ring is free
  {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* stg = (bf16_t*)attn_smem + wave * 4096;
#pragma unroll
    for (int db = 0; db < NOB; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (short)f2bf(oacc[db][4 * j + r] * inv);
        *(bf16x4*)(stg + ql * 128 + (((db * 8 + 2 * j + hh) ^ ql) << 2)) = o;
      }
    if (p.lse && hh == 0 && qrow < p.Sq) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = (l > 0.f) ? m + log2f(l) : -1e30f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private slice: own writes visible to own reads
    const int rr = lane >> 4, ch = lane & 15;
    bf16_t* ob = p.o + (long)b * p.o_bs + (long)h * D + ch * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 4 + rr;
      const bf16x4 lo = *(const bf16x4*)(stg + r * 128 + (((2 * ch) ^ r) << 2));
      const bf16x4 hi = *(const bf16x4*)(stg + r * 128 + (((2 * ch + 1) ^ r) << 2));
      if (qw0 + r < p.Sq && ch * 8 < D)
        *(bf16x8*)(ob + (long)(qw0 + r) * p.o_ts) = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  }
}
