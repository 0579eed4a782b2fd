// D = 128 / 96 attention backward, round 5: ONE wave per SIMD (the whole 512-register file), 32x32x16 MFMAs, 64 rows per wave.
// Included by attention.hip behind attention_d128.h (one translation unit; the launchers live there).
//
// Why: the round-3/4 kernels (attention_d128.h: 4 waves x 32 rows, 16x16x32 MFMAs, two blocks per CU) read 1 KB of LDS
// fragments per 8 K MACs from eight waves — LDS fragment time equals matrix-pipe time, measured 39-44 % matrix-pipe utilisation — and every
// restructuring through the compiler died on register pressure.  Here:
//   * 32x32x16 MFMAs: a 1 KB fragment feeds 16 K MACs, and with 64 rows per wave every LDS fragment feeds TWO of them: 0.5 KB of LDS reads per
//     MFMA (the 4-wave GEMM's ratio), a quarter of the old kernels' LDS traffic per flop;
//   * REGISTERS ARE ASSIGNED BY HAND.  The compiler is fenced into v0..v63 (`amdgpu_num_vgpr(64)`; -amdgpu-spill-vgpr-to-agpr=0 in the Makefile) for
//     addresses, loop control and the prologue / epilogue; v64..v255 and a0..a255 are named literally in the asm statements (the maps are at the top
//     of each kernel): the 256 accumulator registers of a wave (dK^T and dV^T of 64 keys; dQ^T of 64 queries + the wave's Q / dO fragments) in
//     AGPRs, score blocks, K fragments, LDS fragment rings and packed operands in VGPRs.  (First version of this file: the same streams with
//     compiler-allocated operands — "+a" accumulator chains, "=v" fragments: 350-800 spilled registers; the allocator reloads B operands from
//     scratch inside the loop and moves accumulator tuples at every control-flow merge.)  tools/audit_asm_owned.py checks the built ISA: no
//     compiler-generated instruction touches a register above v63 or any AGPR.
//   * every MFMA / LDS read / LDS-DMA piece / softmax instruction is an `asm volatile` statement in program order, so the instruction stream is
//     placed by hand: <= 5 non-MFMA issues per 32-cycle MFMA slot, reads two to three operands ahead on ONE in-order stream with counted
//     lgkmcnt / vmcnt, one barrier per tile;
//   * the second GEMM's operand is the first GEMM's accumulator: 8 consecutive registers of a 32x32 C block are one bf16x8 B operand whose k-slots
//     are tokens 16 t + 8 (s >> 2) + 4 hh + (s & 3); the other operand gathers the SAME tokens with two transposing reads (4 consecutive tokens
//     each), as in the round-3 forward kernel;
//   * the per-query statistics ride in the MFMA's C operand: dK/dV reads -lse / c and -delta from LDS straight INTO the score blocks, so
//     S' = Q K^T - lse / c, P = exp2(c S'), dS = P dP' cost 4 VALU per element instead of 6; dQ keeps -delta in a constant C block.
// LDS tile layout (every streamed 32-token tile): token r at (r >> 2) * 1040 + (r & 3) * 256 bytes (a 1 KB LDS-DMA piece holds 4 tokens; 16 bytes
// of padding per piece), 16-byte chunk c of a token at position c ^ ((r & 3) << 2).  Conflict-free for BOTH access kinds: the row-wise
// ds_read_b128 (16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}: the padding rotates the four token groups onto different bank quarters, the
// XOR spreads the four tokens of a group) and the transposing ds_read_b64_tr_b16 (32 lanes = 4 tokens x 4 chunks x 2 halves).  All fragment
// addresses are one loop-invariant VGPR + a 16-bit immediate.
// Numerics: same formulas and rounding points as the round-3 kernels (P, dS rounded to bf16 as MFMA operands, fp32 accumulation, delta = rowsum(dO o O)
// in fp32, dq / dk scaled by `scale` and rounded once; fused RoPE^T as store_row128<true>): results agree to fp32 summation order.
#pragma once
#include "attention_d128.h"

constexpr int B64_PIECE = 1040;                       // bytes: 4 tokens x 256 B + 16 B of padding
constexpr int B64_TILE = 8 * B64_PIECE;               // 32 tokens
constexpr int B64_DQ_STAGE = 2 * B64_TILE;            // K tile | V tile
constexpr int B64_DQ_RING = 4 * B64_DQ_STAGE;          // 66560 bytes
constexpr int B64_DQ_LDS = B64_DQ_RING + 4 * 8192;       // + one 8 KB staging slice per wave for the whole-line dQ stores
constexpr int B64_KV_STAGE = 2 * B64_TILE + 256;      // Q tile | dO tile | 32 lse' | 32 delta'
constexpr int B64_KV_RING = 4 * B64_KV_STAGE;         // 67584 bytes
constexpr int B64_KV_LDS = B64_KV_RING + 64 * B64_PIECE;      // + the block's 256 V rows (66560 bytes): 134144

// Every register the kernels below name literally.  `amdgpu_num_vgpr(64)` is only a budget: under pressure the allocator goes past it (first GPU run of
// the epilogue's table prefetch: two address registers of the epilogue lived in v96 / v97 across the loop -> memory fault).  B64_FENCE() is an empty
// asm statement that CLOBBERS all of them: no compiler value that is live across a fence can be allocated there.  One fence sits in front of the first
// literal write, one at the head of every loop iteration, one behind the loops; tools/audit_asm_owned.py checks the built ISA.
#define B64_OWNED \
  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", \
  "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", \
  "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", \
  "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", \
  "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", \
  "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", \
  "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", \
  "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", \
  "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", \
  "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", \
  "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", \
  "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", \
  "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", \
  "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", \
  "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
  "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", \
  "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", \
  "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", \
  "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", \
  "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", \
  "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", \
  "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", \
  "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define B64_FENCE() asm volatile("" ::: B64_OWNED)
#define B64_OWNED_A \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", \
  "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", \
  "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", \
  "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", \
  "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", \
  "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", \
  "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", \
  "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", \
  "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", \
  "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", \
  "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", \
  "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", \
  "a249", "a250", "a251", "a252", "a253", "a254", "a255"

// ---- asm statements on literal registers: register numbers are "n" operands printed with %c
// B64_ABL (dev aid, tools/attn_bwd64_ablate.sh; results are wrong): 1 = no softmax arithmetic, 2 = no LDS fragment reads, 4 = no LDS-DMA / barriers
#ifndef B64_ABL
#define B64_ABL 0
#endif
#if B64_ABL & 2
#define R_RD128(R, ADDR, OFF) asm volatile("" ::"v"(ADDR))
#define R_RDTR(R, ADDR, OFF) asm volatile("" ::"v"(ADDR))
#else
#define R_RD128(R, ADDR, OFF) asm volatile("ds_read_b128 v[%c1:%c2], %0 offset:%c3" ::"v"(ADDR), "n"(R), "n"((R) + 3), "n"(OFF))
#define R_RDTR(R, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 v[%c1:%c2], %0 offset:%c3" ::"v"(ADDR), "n"(R), "n"((R) + 1), "n"(OFF))
#endif
// D (= C) class, A class, B class as string literals "v" / "a"
#define R_MF(DC, AC, BC, D0, A0, B0)                                                                                                   \
  asm volatile("v_mfma_f32_32x32x16_bf16 " DC "[%c0:%c1], " AC "[%c2:%c3], " BC "[%c4:%c5], " DC "[%c0:%c1]" ::"n"(D0), "n"((D0) + 15), \
               "n"(A0), "n"((A0) + 3), "n"(B0), "n"((B0) + 3))
#define R_MF_Z(DC, AC, BC, D0, A0, B0)                                                                                                 \
  asm volatile("v_mfma_f32_32x32x16_bf16 " DC "[%c0:%c1], " AC "[%c2:%c3], " BC "[%c4:%c5], 0" ::"n"(D0), "n"((D0) + 15), "n"(A0),     \
               "n"((A0) + 3), "n"(B0), "n"((B0) + 3))
#define R_MF_C(DC, AC, BC, D0, A0, B0, C0)                                                                                             \
  asm volatile("v_mfma_f32_32x32x16_bf16 " DC "[%c0:%c1], " AC "[%c2:%c3], " BC "[%c4:%c5], " DC "[%c6:%c7]" ::"n"(D0), "n"((D0) + 15), \
               "n"(A0), "n"((A0) + 3), "n"(B0), "n"((B0) + 3), "n"(C0), "n"((C0) + 15))
#if B64_ABL & 1
#define R_FMA(R, A, B) asm volatile("" ::"v"(A), "v"(B))
#define R_MULV(R, A) asm volatile("" ::"v"(A))
#define R_MULR(R, R2)
#define R_EXP(R)
#define R_CVT(RD, R0, R1)
#else
#define R_FMA(R, A, B) asm volatile("v_fma_f32 v%c0, v%c0, %1, %2" ::"n"(R), "v"(A), "v"(B))        /* vR = vR * A + B (compiler values) */
#define R_MULV(R, A) asm volatile("v_mul_f32 v%c0, v%c0, %1" ::"n"(R), "v"(A))
#define R_MULR(R, R2) asm volatile("v_mul_f32 v%c0, v%c0, v%c1" ::"n"(R), "n"(R2))
#define R_EXP(R) asm volatile("v_exp_f32 v%c0, v%c0" ::"n"(R))
#define R_CVT(RD, R0, R1) asm volatile("v_cvt_pk_bf16_f32 v%c0, v%c1, v%c2" ::"n"(RD), "n"(R0), "n"(R1))
#endif
#define R_FMAL(R, A, RL) asm volatile("v_fma_f32 v%c0, v%c0, %1, v%c2" ::"n"(R), "v"(A), "n"(RL))     /* vR = vR * A + v[RL] (a literal register) */
#define R_VMOV(R, X) asm volatile("v_mov_b32 v%c0, %1" ::"n"(R), "v"(X))
#define R_AWRITE(R, X) asm volatile("v_accvgpr_write_b32 a%c0, %1" ::"n"(R), "v"(X))
#define R_AZERO(R) asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(R))
// keep vR where LO <= E < HI (E = the element's inline-constant offset, LO / HI / NI = compiler VGPRs; NI holds -inf), else -inf
#define R_MASK(R, E, LO, HI, NI)                                                                                                       \
  asm volatile("v_cmp_ge_i32 vcc, %c4, %1\n\tv_cndmask_b32 v%c0, %3, v%c0, vcc\n\tv_cmp_lt_i32 vcc, %c4, %2\n\tv_cndmask_b32 v%c0, %3, v%c0, vcc" \
               ::"n"(R), "v"(LO), "v"(HI), "v"(NI), "n"(E) : "vcc")
#define B64_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N))
#if B64_ABL & 4
#define B64_VMCNT(N)
#define B64_BAR()
#define B64_DMA16(M0, VOFF, RS, SOFF) asm volatile("" ::"s"(M0), "v"(VOFF), "s"(RS), "s"(SOFF))
#define B64_DMA4(M0, VOFF, RS, SOFF) asm volatile("" ::"s"(M0), "v"(VOFF), "s"(RS), "s"(SOFF))
#else
#define B64_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define B64_BAR() asm volatile("s_barrier" ::: "memory")
#define B64_DMA16(M0, VOFF, RS, SOFF) ATTN_DMA16(M0, VOFF, RS, SOFF)
#define B64_DMA4(M0, VOFF, RS, SOFF) ATTN_DMA4(M0, VOFF, RS, SOFF)
#endif

// A generic lambda only captures what it names OUTSIDE dependent `if constexpr` branches (clang resolves the implicit captures at definition time):
// every stream lambda below starts by naming what its branches use.
template <class... T>
static __device__ __forceinline__ void b64_use(T&...) {}

// 4 consecutive accumulator registers (a[R .. R + 3]) -> 4 floats
template <int R>
static __device__ __forceinline__ f32x4 b64_aread() {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3)
               : B64_OWNED);                // (no compiler value may sit in an asm-owned register across an accumulator read: see B64_FENCE)
  return v;
}
// one token's D features held as 32x32 C-layout accumulator registers (a[A0 + 16 db + i]: feature 32 db + 8 (i >> 2) + 4 hh + (i & 3) of the lane's token)
// -> bf16 in global memory, optionally rotated back by RoPE^T (same rounding points as store_row128<true>: x1 = feature f < D/2, x2 = f + D/2; both
// sit in the same lane: D = 128 blocks db and db + 2; D = 96: 8-feature groups u and u + 6).  Four registers at a time: the compiler's share of
// the register file is small (v0..v63).
// Epilogue.  The RoPE tables of the wave's two rows are fetched by asm buffer loads into the (by then dead) asm-owned VGPRs v[64 + 64 r ..] BEFORE the
// accumulators are read: one latency for the whole epilogue (with the loads inside the store loop it was a chain of 16 dependent global round trips,
// ~20 us of a ~50 us block; as compiler values the 64 table registers do not fit beside its 64-register budget).  Row r: cos quads at 64 + 64 r + 4 u,
// sin quads at 96 + 64 r + 4 u (u = 8-feature group of the first half: table entries 8 u + 4 hh .. + 3).
template <int NOB>
static __device__ __forceinline__ void b64_rope_issue(const float* cs, const float* sn, const uint32_t (&voff)[2]) {
  const fwdm_u32x4s rsC = attn_make_rs(cs, 0x7fffffff), rsS = attn_make_rs(sn, 0x7fffffff);
  B64_FENCE();                                // voff (and anything else live here) cannot sit in v64..v191: the loads below overwrite them
  asm volatile("s_nop 4" ::: "memory");
  vp_static_for<2 * 2 * NOB>([&](auto i_) __attribute__((always_inline)) {
    constexpr int r = decltype(i_)::value / (2 * NOB), u = decltype(i_)::value % (2 * NOB);
    b64_use(voff, rsC, rsS);
    asm volatile("buffer_load_dwordx4 v[%c2:%c3], %0, %1, 0 offen offset:%c4" ::"v"(voff[r]), "s"(rsC), "n"(64 + 64 * r + 4 * u), "n"(64 + 64 * r + 4 * u + 3),
                 "n"(32 * u) : "memory");
    asm volatile("buffer_load_dwordx4 v[%c2:%c3], %0, %1, 0 offen offset:%c4" ::"v"(voff[r]), "s"(rsS), "n"(96 + 64 * r + 4 * u), "n"(96 + 64 * r + 4 * u + 3),
                 "n"(32 * u) : "memory");
  });
}
template <int R>
static __device__ __forceinline__ f32x4 b64_vread() {                 // v[R .. R + 3] (asm-owned) -> compiler values; fenced like b64_aread
  f32x4 v;
  asm volatile("v_mov_b32 %0, v%c4\n\tv_mov_b32 %1, v%c5\n\tv_mov_b32 %2, v%c6\n\tv_mov_b32 %3, v%c7"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3) : B64_OWNED);
  return v;
}
// LDS = true: the row goes into a wave-private staging slice instead (32 rows x 128 features, 8-byte unit w of row r at position w ^ r: conflict-free
// for these 16-lane write groups and for b64_flush_rows' row-wise reads); dst = the slice, ql = the lane's row in it.  A row-per-lane global store
// (32 rows x 8 bytes per instruction) moves ~7 B/clk per CU, whole cache lines 53 (tools/probes/store_pattern_probe.hip, the forward's store path).
template <bool ROPE, int NOB, int A0, int ROW, bool LDS = false>
static __device__ __forceinline__ void b64_store_row(bf16_t* dst, float scale, int hh, int ql = 0) {
  constexpr int NG = NOB * 4;                           // 8-feature groups; group u = registers A0 + 4 u .. + 3 = features 8 u + 4 hh + e
  auto put = [&](int f, bf16x4 v) __attribute__((always_inline)) {
    if constexpr (LDS) *(bf16x4*)(dst + ql * 128 + (((f >> 2) ^ ql) << 2)) = v; else *(bf16x4*)(dst + f) = v;
  };
  if constexpr (!ROPE) {
    vp_static_for<NG>([&](auto u_) __attribute__((always_inline)) {
      constexpr int u = decltype(u_)::value;
      const f32x4 x = b64_aread<A0 + 4 * u>();
      bf16x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = (short)f2bf(x[e] * scale);
      put(8 * u + 4 * hh, a);
    });
  } else {
    vp_static_for<NG / 2>([&](auto u_) __attribute__((always_inline)) {
      constexpr int u = decltype(u_)::value;
      const f32x4 xa = b64_aread<A0 + 4 * u>(), xb = b64_aread<A0 + 4 * (u + NG / 2)>();
      const f32x4 c4 = b64_vread<64 + 64 * ROW + 4 * u>(), s4 = b64_vread<96 + 64 * ROW + 4 * u>();
      const int f = 8 * u + 4 * hh;
      bf16x4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x1 = bfround(xa[e] * scale), x2 = bfround(xb[e] * scale);
        const float ss = -s4[e];
        a[e] = (short)f2bf(bfround(x1 * c4[e]) + bfround(-x2 * ss));
        b[e] = (short)f2bf(bfround(x2 * c4[e]) + bfround(x1 * ss));
      }
      put(f, a);
      put(4 * NG + f, b);                               // + D / 2 features
    });
  }
}

// a staged 32-row block -> global memory as WHOLE rows: 4 rows x 16 bytes per lane per instruction.  gbase = row 0 / feature 0 of the block,
// ts = the row stride in elements, rows_valid = rows of the block that exist (may be <= 0)
template <int D>
static __device__ __forceinline__ void b64_flush_rows(const bf16_t* stg, bf16_t* gbase, long ts, int rows_valid, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private slice: own writes visible to own reads
  const int rr = lane >> 4, ch = lane & 15;
#pragma unroll
  for (int t4 = 0; t4 < 8; ++t4) {
    const int r = t4 * 4 + rr;
    const bf16x4 lo = *(const bf16x4*)(stg + r * 128 + (((2 * ch) ^ r) << 2));
    const bf16x4 hi = *(const bf16x4*)(stg + r * 128 + (((2 * ch + 1) ^ r) << 2));
    if (r < rows_valid && ch * 8 < D) *(bf16x8*)(gbase + (long)r * ts + ch * 8) = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the slice is rewritten by the next block)
}

// ================================================================================================
// dQ.  Block = 4 waves x 64 queries of one q head; K / V stream through a 4-stage ring of 32-key tiles.  Swapped products: S^T = K Q^T and
// dP^T = V dO^T (lane = query: lse / delta are lane scalars), dQ^T += K^T dS^T.  Per tile and wave 48 MFMAs in three segments of 16:
//   A  dP^T of tile i            under  P = exp2(c S^T - lse) of tile i (S^T was computed one iteration earlier)
//   B  S^T of tile i + 1         under  dS = P dP' and its bf16 packing
//   C  dQ^T += K^T dS^T          under  the transposing reads and the LDS-DMA pieces of tile i + 3
// This kernel also computes delta = rowsum(dO o O) and writes the (-lse / c, -delta) planes the dK/dV kernel streams: launched first.
// Register map (D = 128; D = 96 uses the same numbers with 6 k-steps / 3 feature blocks):
//   a[0:127]   dQ^T accumulators: (q block qb, feature block db) at 64 qb + 16 db
//   a[128:191] Q fragments (qb, ks) at 128 + 32 qb + 4 ks;  a[192:255] dO fragments at 192 + 32 qb + 4 ks
//   v[64:95] S^T of even tiles (16 qb), v[96:127] of odd tiles, v[128:159] dP^T, v[160:191] the constant -delta blocks
//   v[192:207] K row fragments (ring of 4), v[208:223] V row fragments, v[224:235] transposed-K operands (ring of 3: lo | hi),
//   v[236:251] packed dS: (qb, t) at 236 + 8 qb + 4 t,  v252 scratch of the mask statement,  v253 / v254 -lse of the two q blocks
// ================================================================================================
template <bool CAUSAL, bool ROPE, int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(64))) void attn_bwd_dq64w_kernel(AttnParams p) {
  static_assert(D == 128 || D == 96, "D");
  constexpr int NKS = D / 16, NOB = D / 32, NOPS = 2 * NOB;
  constexpr int A_ACC = 0, A_QF = 128, A_DOF = 192, V_SE = 64, V_SO = 96, V_DP = 128, V_ND = 160, V_KFR = 192, V_VFR = 208, V_TR = 224, V_PK = 236, V_NL = 253;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  asm volatile("" ::: "v255", "a255");                   // the wave owns its SIMD's whole register file (see gemm_nt_256w4)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  const int nqb = (p.Sq + 255) >> 8;
  const int qblk = nqb - 1 - VP_BZ(p);                 // z is the slowest dispatch index: heavy (late) causal blocks first
  const int hx = blockIdx.x;
  const int h = (p.Hq & 7) == 0 ? (hx & 7) * (p.Hq >> 3) + (hx >> 3) : hx;     // whole GQA groups per XCD
  const int b = VP_BY(p), hk = h / (p.Hq / p.Hkv);
  const int q0 = qblk * 256, qw0 = q0 + wave * 64;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq, Sq = p.Sq, window = p.window;
  const float c = p.scale * LOG2E;
  const long nrows = (long)p.B * p.Hq * p.Sq;

  // ---- Q / dO fragments (B operands: lane = query, features 16 ks + 8 hh .. + 7) straight into their AGPRs (buffer loads with an AGPR destination:
  // no pressure on the compiler's 64 VGPRs), delta = rowsum(dO o O) from the loaded dO fragments, -lse; the two statistics planes
  float nlse[2], npart[2];
  {
    const fwdm_u32x4s rsQh = attn_make_rs(p.q + (long)b * p.q_bs + (long)h * D, (((long)Sq - 1) * p.q_ts + D) * 2);
    const fwdm_u32x4s rsGh = attn_make_rs(p.dout + (long)b * p.do_bs + (long)h * D, (((long)Sq - 1) * p.do_ts + D) * 2);
    uint32_t voq[2], vog[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrc = min(qw0 + 32 * qb + ql, Sq - 1);  // clamped (unconditional loads); rows >= Sq are never stored
      voq[qb] = (uint32_t)qrc * (uint32_t)(p.q_ts * 2) + 16u * (uint32_t)hh;
      vog[qb] = (uint32_t)qrc * (uint32_t)(p.do_ts * 2) + 16u * (uint32_t)hh;
    }
    asm volatile("s_nop 4" ::: "memory");               // (descriptor SGPRs fresh from v_readfirstlane)
    vp_static_for<2 * NKS>([&](auto i_) __attribute__((always_inline)) {
      constexpr int qb = decltype(i_)::value / NKS, ks = decltype(i_)::value % NKS;
      b64_use(voq, vog, rsQh, rsGh);
      asm volatile("buffer_load_dwordx4 a[%c2:%c3], %0, %1, 0 offen offset:%c4" ::"v"(voq[qb]), "s"(rsQh), "n"(A_QF + 32 * qb + 4 * ks),
                   "n"(A_QF + 32 * qb + 4 * ks + 3), "n"(32 * ks) : "memory");
      asm volatile("buffer_load_dwordx4 a[%c2:%c3], %0, %1, 0 offen offset:%c4" ::"v"(vog[qb]), "s"(rsGh), "n"(A_DOF + 32 * qb + 4 * ks),
                   "n"(A_DOF + 32 * qb + 4 * ks + 3), "n"(32 * ks) : "memory");
    });
  }
  // the O rows (compiler loads, one q block at a time: 32 registers) fly together with the fragment loads; one wait for everything
  vp_static_for<2>([&](auto qb_) __attribute__((always_inline)) {
    constexpr int qb = decltype(qb_)::value;
    const int qrow = qw0 + 32 * qb + ql;
    const int qrc = min(qrow, Sq - 1);
    const bf16_t* op_ = p.o + (long)b * p.o_bs + (long)qrc * p.o_ts + (long)h * D;
    const long sidx = ((long)b * p.Hq + h) * Sq + qrc;
    bf16x8 ov[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) ov[ks] = *(const bf16x8*)(op_ + ks * 16 + hh * 8);
    const float lse = p.lse[sidx];
    B64_VMCNT(0);
    float part = 0.f;
    vp_static_for<NKS>([&](auto ks_) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_)::value;
      b64_use(ov, part);
      const bf16x8 g8 = __builtin_bit_cast(bf16x8, b64_aread<A_DOF + 32 * qb + 4 * ks>());
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(bf2f((bf16_t)g8[e]), bf2f((bf16_t)ov[ks][e]), part);
    });
    part += __shfl_xor(part, 32, 64);
    nlse[qb] = -lse;
    npart[qb] = -part;
    if (hh == 0 && qrow < Sq) {
      p.delta[sidx] = -lse / c;                        // plane 0: -lse / c   (S' = S - lse / c, P = exp2(c S'))
      p.delta[nrows + sidx] = -part;                   // plane 1: -delta
    }
  });
  vp_static_for<2 * NOB * 16>([&](auto i_) __attribute__((always_inline)) {
    constexpr int i = decltype(i_)::value;
    R_AZERO(A_ACC + 64 * (i / (NOB * 16)) + (i % (NOB * 16)));
  });

  int kend = kvlen;
  if (CAUSAL) kend = min(kend, q0 + 256 + off);
  int kstart = 0;
  if (window > 0) kstart = max(0, (q0 + off - window + 1)) & ~31;
  const int nit = (B64_ABL & 8) ? 0 : (kend > kstart ? (kend - kstart + 31) / 32 : 0);      // (ablation 8: prologue + epilogue only)
  // tiles this WAVE computes: [first_w, last_w]; the others only keep the block's barrier / DMA cadence
  int last_w = CAUSAL ? min(nit - 1, (qw0 + 63 + off - kstart) >> 5) : nit - 1;
  int first_w = 0;
  if (window > 0) first_w = max(0, (qw0 + off - window + 1 - kstart) >> 5);
  if (last_w < first_w || qw0 >= Sq) { last_w = -1; first_w = nit; }
  // The ring stage is a literal of the four-way unrolled loop, so the wave enters it at a multiple of four: up to three tiles below a sliding window's
  // lower edge are computed fully masked (P = 0, dS = 0: they add nothing) instead of entering the loop at an arbitrary phase — the goto-entered loop
  // was irreducible and the compiler's dispatcher for it cost ~25 scalar instructions and several branches in EVERY iteration of every launch.
  else first_w &= ~3;

  // ---- LDS-DMA: K / V pieces through buffer descriptors of this (batch, kv head); rows past Skv read as zeros
  const fwdm_u32x4s rsK = attn_make_rs(p.k + (long)b * p.k_bs + (long)hk * D, (((long)p.Skv - 1) * p.k_ts + D) * 2);
  const fwdm_u32x4s rsV = attn_make_rs(p.v + (long)b * p.v_bs + (long)hk * D, (((long)p.Skv - 1) * p.v_ts + D) * 2);
  const uint32_t ldsb = attn_lds_addr(attn_smem);
  const uint32_t kts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.k_ts * 2)), vts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.v_ts * 2));
  uint32_t vK, vV;
  {
    const int da = lane >> 4, dch = (lane & 15) ^ (da << 2);
    vK = (uint32_t)da * kts2 + (uint32_t)dch * 16u;
    vV = (uint32_t)da * vts2 + (uint32_t)dch * 16u;
    asm volatile("" : "+v"(vK), "+v"(vV));
  }
  const uint32_t m0w = __builtin_amdgcn_readfirstlane(ldsb + (uint32_t)wave * (uint32_t)B64_PIECE);
  // piece PC of tile T into stage ST: 0 / 1 = K token groups wave, wave + 4; 2 / 3 = V likewise
#define DQ_DMA(T, ST, PC)                                                                                       \
  {                                                                                                             \
    const uint32_t row_ = (uint32_t)(kstart + min((T), nit - 1) * 32 + 4 * wave + (((PC) & 1) ? 16 : 0));       \
    const uint32_t so_ = row_ * (((PC) & 2) ? vts2 : kts2);                                                     \
    const uint32_t m0_ = m0w + (uint32_t)((ST) * B64_DQ_STAGE + (((PC) & 1) ? 4 * B64_PIECE : 0) + (((PC) & 2) ? B64_TILE : 0)); \
    if ((PC) & 2) B64_DMA16(m0_, vV, rsV, so_); else B64_DMA16(m0_, vK, rsK, so_);                            \
  }
#define DQ_DMA4(T, ST) { DQ_DMA(T, ST, 0) DQ_DMA(T, ST, 1) DQ_DMA(T, ST, 2) DQ_DMA(T, ST, 3) }

  // ---- fragment addresses (loop-invariant; stage / tensor / k-step in the immediate)
  uint32_t rowa[4], tra[NOB];
  {
    const int rj = ql >> 2, ra = ql & 3;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      rowa[m] = ldsb + (uint32_t)(rj * B64_PIECE + ra * 256 + 16 * hh + 64 * (m ^ ra));
      asm volatile("" : "+v"(rowa[m]));
    }
    const int fr = lane & 15, ta = fr >> 2, tx = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), th = lane & 1;
#pragma unroll
    for (int db = 0; db < NOB; ++db) {
      tra[db] = ldsb + (uint32_t)(hh * B64_PIECE + ta * 256 + 64 * (db ^ ta) + 16 * tx + 8 * th);
      asm volatile("" : "+v"(tra[db]));
    }
  }
  float cc = c, ninf = -INFINITY;
  asm volatile("" : "+v"(cc), "+v"(ninf));

  // LDS reads form ONE in-order stream per wave that never drains inside the loop: every k-step / operand step issues a fixed number of reads and
  // waits with the same counted lgkmcnt.  Stream of an iteration (tile i in stage ST):
  //   A: [V rows 3..NKS-1 of tile i | K rows 0..2 of tile i + 1]          one read per k-step, lgkmcnt(3): V row ks landed
  //   B: [K rows 3..NKS-1 of tile i + 1 | op0.lo op0.hi op1.lo]           one read per k-step, lgkmcnt(3)
  //   C: [op1.hi op2.lo ... op(N-1).hi | V rows 0..2 of tile i + 1]       two reads per operand step, lgkmcnt(3): operand n landed
#define DQ_KRD(KS, ST) R_RD128(V_KFR + 4 * ((KS) & 3), rowa[(KS) >> 1], (ST) * B64_DQ_STAGE + 32 * ((KS) & 1))
#define DQ_VRD(KS, ST) R_RD128(V_VFR + 4 * ((KS) & 3), rowa[(KS) >> 1], (ST) * B64_DQ_STAGE + B64_TILE + 32 * ((KS) & 1))
#define DQ_TRLO(N, ST) R_RDTR(V_TR + 4 * ((N) % 3), tra[(N) % NOB], (ST) * B64_DQ_STAGE + ((N) / NOB) * 4 * B64_PIECE)
#define DQ_TRHI(N, ST) R_RDTR(V_TR + 4 * ((N) % 3) + 2, tra[(N) % NOB], (ST) * B64_DQ_STAGE + ((N) / NOB) * 4 * B64_PIECE + 2 * B64_PIECE)
  // element G of segment C's read stream (G = 0 is op1.hi)
#define DQ_CSTREAM(G, ST)                                                                                       \
  {                                                                                                             \
    if constexpr ((G) < 2 * NOPS - 3) {                                                                         \
      if constexpr (((G) & 1) == 0) { DQ_TRHI(((G) + 3) / 2, ST); } else { DQ_TRLO(((G) + 3) / 2, ST); }        \
    } else { DQ_VRD((G) - (2 * NOPS - 3), ((ST) + 1) & 3); }                                                    \
  }

  if constexpr (ROPE) asm volatile("" ::"s"(p.rope_cos), "s"(p.rope_sin), "s"(p.rope_pos));      // (the epilogue's kernel arguments: loaded here, not inside the stream)
  // from here on v64..v255 are asm-owned (tools/audit_asm_owned.py): the constant -delta blocks first
  B64_FENCE();
  vp_static_for<32>([&](auto i_) __attribute__((always_inline)) {
    constexpr int i = decltype(i_)::value;
    b64_use(npart);
    R_VMOV(V_ND + i, npart[i >> 4]);
  });
  // -lse of the wave's two q blocks in literal registers as well: as compiler values they were spilled under the prologue's pressure and reloaded
  // from scratch at the loop's entry (behind the first LDS-DMA pieces: a vmcnt(0) that drains the ring once per block)
  R_VMOV(V_NL, nlse[0]);
  R_VMOV(V_NL + 1, nlse[1]);
  if (nit > 0) {
    DQ_DMA4(0, 0) DQ_DMA4(1, 1) DQ_DMA4(2, 2)
    B64_VMCNT(8);                                       // tile 0 landed (this wave's part)
    B64_BAR();
  }
  // head: tiles below this wave's window only keep the cadence (barrier i: tile i + 1 landed for everybody, stage (i - 1) & 3 free)
  for (int it = 0; it < min(first_w, nit); ++it) {
    B64_FENCE();
    B64_VMCNT(4);
    B64_BAR();
    DQ_DMA4(it + 3, (it + 3) & 3)
  }
  B64_FENCE();
  // S^T of the wave's first tile and the first V rows (a run-time stage only here -> one copy per stage)
#define DQ_FIRST(SN, ST)                                                                                        \
  {                                                                                                             \
    DQ_KRD(0, ST); DQ_KRD(1, ST); DQ_KRD(2, ST);                                                                \
    vp_static_for<NKS>([&](auto ks_) __attribute__((always_inline)) {                                           \
      constexpr int ks = decltype(ks_)::value;                                                                  \
      b64_use(rowa);                                                                                            \
      if constexpr (ks + 3 < NKS) { DQ_KRD(ks + 3, ST); } else { DQ_VRD(ks + 3 - NKS, ST); }                    \
      B64_LGKM(3);                                                                                              \
      if constexpr (ks == 0) {                                                                                  \
        R_MF_Z("v", "v", "a", SN, V_KFR, A_QF); R_MF_Z("v", "v", "a", SN + 16, V_KFR, A_QF + 32);               \
      } else {                                                                                                  \
        R_MF("v", "v", "a", SN, V_KFR + 4 * (ks & 3), A_QF + 4 * ks); R_MF("v", "v", "a", SN + 16, V_KFR + 4 * (ks & 3), A_QF + 32 + 4 * ks); \
      }                                                                                                         \
    });                                                                                                         \
  }
  B64_LGKM(0);                                          // (no scalar load of the prologue may be in flight inside the counted-lgkmcnt stream)
  if (last_w >= 0) DQ_FIRST(V_SE, 0)

  // exponentials of elements 2 X, 2 X + 1 of the tile's 32 (element e = register e & 15 of q block e >> 4: v[SC + e])
#define DQ_EXPV(SC, X)                                                                                          \
  {                                                                                                             \
    R_FMAL(SC + 2 * (X), cc, V_NL + ((2 * (X)) >> 4));                                                          \
    R_FMAL(SC + 2 * (X) + 1, cc, V_NL + ((2 * (X) + 1) >> 4));                                                  \
    R_EXP(SC + 2 * (X));                                                                                        \
    R_EXP(SC + 2 * (X) + 1);                                                                                    \
  }
  // dS = P dP' and its bf16 packing under the S^T MFMAs of the next tile: MPM multiplies per MFMA from MFMA 1 on (element e under MFMA 1 + e / MPM),
  // two packs per MFMA from MFMA P0 on (pack n = elements 2n, 2n + 1, under MFMA P0 + n / 2, always behind its multiplies; pack n is word n of
  // v[V_PK ..]: (qb, t, word) = (n >> 3, (n >> 2) & 1, n & 3)).
  // (dP's last MFMA is >= 12 issue states old when its first element is read: 4 riders + wait + barrier + the stream reads sit in between.)
#define DQ_DSV(SC, M, MPM, P0)                                                                                  \
  {                                                                                                             \
    vp_static_for<MPM>([&](auto j_) __attribute__((always_inline)) {                                            \
      constexpr int e = (MPM) * ((M) - 1) + decltype(j_)::value;                                                \
      if constexpr ((M) >= 1 && e < 32) R_MULR(V_DP + e, SC + e);                                               \
    });                                                                                                         \
    vp_static_for<2>([&](auto j_) __attribute__((always_inline)) {                                              \
      constexpr int n = 2 * ((M) - (P0)) + decltype(j_)::value;                                                 \
      if constexpr ((M) >= (P0) && n < 16) R_CVT(V_PK + n, V_DP + 2 * n, V_DP + 2 * n + 1);                     \
    });                                                                                                         \
  }
  // One iteration.  ST = the tile's stage (literal), SC = its S^T block, SN = the next tile's.
#define DQ_ITER(IT, ST, SC, SN)                                                                                 \
  {                                                                                                             \
    B64_FENCE();                                                                                                \
    const int k0_ = kstart + (IT) * 32;                                                                         \
    const bool need_mask = (k0_ + 32 > kvlen) || (CAUSAL && (k0_ + 31 > qw0 + off)) || (window > 0 && k0_ <= qw0 + 63 + off - window); \
    if (need_mask) {                 /* register r of q block qb = key k0 + e(r) + 4 hh, e(r) = (r & 3) + 8 (r >> 2): keep lo <= e < hi */ \
      int lm_ = threadIdx.x & 63;    /* (lane-derived values from an opaque lane id: nothing lane-dependent stays live across the loop) */ \
      asm volatile("" : "+v"(lm_));                                                                             \
      const int qlm_ = lm_ & 31, hhm_ = lm_ >> 5;                                                               \
      vp_static_for<2>([&](auto qb_) __attribute__((always_inline)) {                                           \
        constexpr int qb = decltype(qb_)::value;                                                                \
        b64_use(ninf);                                                                                          \
        const int qrow_ = qw0 + 32 * qb + qlm_;                                                                 \
        const int hi_ = CAUSAL ? min(kvlen - k0_ - 4 * hhm_, qrow_ + off - k0_ - 4 * hhm_ + 1) : kvlen - k0_ - 4 * hhm_; \
        const int lo_ = window > 0 ? qrow_ + off - window - k0_ - 4 * hhm_ + 1 : -(1 << 30);                    \
        vp_static_for<16>([&](auto r_) __attribute__((always_inline)) {                                         \
          constexpr int r = decltype(r_)::value;                                                                \
          b64_use(lo_, hi_, ninf);                                                                              \
          R_MASK(SC + 16 * qb + r, (r & 3) + 8 * (r >> 2), lo_, hi_, ninf);                                     \
        });                                                                                                     \
      });                                                                                                       \
    }                                                                                                           \
    /* ---- A: dP^T = V dO^T (C = -delta) under this tile's exponentials; barrier in the middle; K rows of the next tile behind it */ \
    vp_static_for<NKS>([&](auto ks_) __attribute__((always_inline)) {                                           \
      constexpr int ks = decltype(ks_)::value;                                                                  \
      b64_use(rowa, cc);                                                                                  \
      if constexpr (ks == NKS - 4) { B64_VMCNT(4); B64_BAR(); }      /* tile IT + 1 landed for everybody; stage (IT - 1) & 3 is free */ \
      if constexpr (ks + 3 < NKS) { DQ_VRD(ks + 3, ST); } else { DQ_KRD(ks + 3 - NKS, ((ST) + 1) & 3); }        \
      B64_LGKM(3);                                                                                              \
      if constexpr (ks == 0) { R_MF_C("v", "v", "a", V_DP, V_VFR, A_DOF, V_ND); }                               \
      else { R_MF("v", "v", "a", V_DP, V_VFR + 4 * (ks & 3), A_DOF + 4 * ks); }                                 \
      DQ_EXPV(SC, 2 * ks)                                                                                       \
      if constexpr (NKS == 6 && ks < 4) { DQ_EXPV(SC, 12 + ks) }                                                \
      if constexpr (ks == 0) { R_MF_C("v", "v", "a", V_DP + 16, V_VFR, A_DOF + 32, V_ND + 16); }                \
      else { R_MF("v", "v", "a", V_DP + 16, V_VFR + 4 * (ks & 3), A_DOF + 32 + 4 * ks); }                       \
      DQ_EXPV(SC, 2 * ks + 1)                                                                                   \
    });                                                                                                         \
    /* ---- B: S^T of the next tile under dS of this one; the first transposing reads behind the K rows */       \
    vp_static_for<NKS>([&](auto ks_) __attribute__((always_inline)) {                                           \
      constexpr int ks = decltype(ks_)::value;                                                                  \
      b64_use(rowa, tra);                                                                                       \
      if constexpr (ks + 3 < NKS) { DQ_KRD(ks + 3, ((ST) + 1) & 3); }                                           \
      else if constexpr (ks + 3 == NKS) { DQ_TRLO(0, ST); }                                                     \
      else if constexpr (ks + 3 == NKS + 1) { DQ_TRHI(0, ST); }                                                 \
      else { DQ_TRLO(1, ST); }                                                                                  \
      B64_LGKM(3);                                                                                              \
      if constexpr (ks == 0) { R_MF_Z("v", "v", "a", SN, V_KFR, A_QF); } else { R_MF("v", "v", "a", SN, V_KFR + 4 * (ks & 3), A_QF + 4 * ks); } \
      DQ_DSV(SC, 2 * ks, (NKS == 8 ? 3 : 4), (NKS == 8 ? 5 : 4))                                                \
      if constexpr (ks == 0) { R_MF_Z("v", "v", "a", SN + 16, V_KFR, A_QF + 32); }                              \
      else { R_MF("v", "v", "a", SN + 16, V_KFR + 4 * (ks & 3), A_QF + 32 + 4 * ks); }                          \
      DQ_DSV(SC, 2 * ks + 1, (NKS == 8 ? 3 : 4), (NKS == 8 ? 5 : 4))                                            \
    });                                                                                                         \
    /* ---- C: dQ^T += K^T dS^T; the LDS-DMA pieces of tile IT + 3; the next tile's first V rows at the end of the stream */ \
    vp_static_for<NOPS>([&](auto n_) __attribute__((always_inline)) {                                           \
      constexpr int n = decltype(n_)::value;                                                                    \
      b64_use(tra, rowa, vK, vV, m0w, kstart, nit, wave, it, rsK, rsV, kts2, vts2);                             \
      DQ_CSTREAM(2 * n, ST) DQ_CSTREAM(2 * n + 1, ST)                                                           \
      B64_LGKM(3);                                                                                              \
      R_MF("a", "v", "v", A_ACC + 16 * (n % NOB), V_TR + 4 * (n % 3), V_PK + 4 * (n / NOB));                    \
      if constexpr (n < 4) DQ_DMA((IT) + 3, ((ST) + 3) & 3, n)                                                  \
      R_MF("a", "v", "v", A_ACC + 64 + 16 * (n % NOB), V_TR + 4 * (n % 3), V_PK + 8 + 4 * (n / NOB));           \
    });                                                                                                         \
  }

  if (last_w >= 0) {
    int it = first_w;                                   // a multiple of four: tile `it` sits in stage 0
    for (;;) {
      DQ_ITER(it, 0, V_SE, V_SO)
      if (++it > last_w) break;
      DQ_ITER(it, 1, V_SO, V_SE)
      if (++it > last_w) break;
      DQ_ITER(it, 2, V_SE, V_SO)
      if (++it > last_w) break;
      DQ_ITER(it, 3, V_SO, V_SE)
      if (++it > last_w) break;
    }
    B64_LGKM(0);                                        // (the stream's last reads: V rows of a tile this wave does not compute)
  }
  // tail: tiles above this wave's diagonal that the block's other waves still need
  B64_FENCE();
  for (int it = max(last_w + 1, min(first_w, nit)); it < nit; ++it) {
    B64_FENCE();
    B64_VMCNT(4);
    B64_BAR();
    DQ_DMA4(it + 3, (it + 3) & 3)
  }
  // epilogue: the RoPE tables of both rows first (they fly while the trailing DMAs drain), then the accumulators.  Every lane-derived value is
  // re-derived HERE from an opaque lane id: kept live across the loops (the compiler hoists address arithmetic) they do not fit its 64 registers
  // and are reloaded from scratch inside the loop (a VMEM load whose wait drains the DMA ring).
  B64_FENCE();
  int ln_ = threadIdx.x & 63;
  asm volatile("" : "+v"(ln_));
  const int ql_ = ln_ & 31, hh_ = ln_ >> 5;
  if constexpr (ROPE) {
    uint32_t vo[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrc = min(qw0 + 32 * qb + ql_, Sq - 1);
      const long pp = (p.rope_pos ? (long)p.rope_pos[(long)b * Sq + qrc] : (long)(qrc + off)) * (D / 2);
      vo[qb] = (uint32_t)(pp * 4 + 16 * hh_);
    }
    b64_rope_issue<NOB>(p.rope_cos, p.rope_sin, vo);
  }
  B64_VMCNT(0);                                         // the tables landed; trailing (dummy) DMAs must not outlive the block's LDS
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the last MFMAs' results are visible to v_accvgpr_read
  {   // whole-line stores through the wave's staging slice behind the ring (one 32-query block at a time)
    bf16_t* stg = (bf16_t*)(attn_smem + B64_DQ_RING) + wave * 4096;
    bf16_t* g0 = p.dq + (long)b * p.dq_bs + (long)qw0 * p.dq_ts + (long)h * D;
    b64_store_row<ROPE, NOB, A_ACC, 0, true>(stg, p.scale, hh_, ql_);
    b64_flush_rows<D>(stg, g0, p.dq_ts, Sq - qw0, ln_);
    b64_store_row<ROPE, NOB, A_ACC + 64, 1, true>(stg, p.scale, hh_, ql_);
    b64_flush_rows<D>(stg, g0 + 32 * (long)p.dq_ts, p.dq_ts, Sq - qw0 - 32, ln_);
  }
#undef DQ_DMA
#undef DQ_DMA4
#undef DQ_KRD
#undef DQ_VRD
#undef DQ_TRLO
#undef DQ_TRHI
#undef DQ_CSTREAM
#undef DQ_FIRST
#undef DQ_EXPV
#undef DQ_DSV
#undef DQ_ITER
}

// ================================================================================================
// dK, dV.  Block = 4 waves x 64 keys of one kv head (256 keys); loops the GQA group's q heads and 32-query tiles, streamed (Q tile | dO tile |
// -lse/c, -delta) through a 4-stage LDS-DMA ring.  Unswapped products: S' = Q K^T + (-lse / c) and dP' = dO V^T + (-delta) (lane = key, registers =
// queries; the per-query statistics are READ FROM LDS INTO the score blocks before the first MFMA accumulates on them), then dV^T += dO^T P,
// dK^T += Q^T dS with P / dS packed from 8 consecutive accumulator registers.  K fragments live in VGPRs, the block's V rows in LDS (the AGPRs hold
// the 256 accumulators).  Per tile and wave 64 MFMAs (D = 128) in four phases of 16; P = exp2(c S') rides under phases 2 and 3, dS = P dP' under
// phase 3, the packs behind them (dS packs in place: words 0..3 of each 8-register half of the dP' block).
// A tile that needs the per-element mask (diagonal, ragged ends, window edge) takes the same stream with a block of compare / select statements
// between phase 1 and 2 that sets masked scores to -inf (P = 0 exactly, dS = 0).
// Register map:  a[0:127] dK^T (key block kb, feature block db) at 64 kb + 16 db;  a[128:255] dV^T likewise
//   v[64:127] K fragments (kb, ks) at 64 + 32 kb + 4 ks;  v[128:159] S' / P (16 kb);  v[160:191] dP' / dS (16 kb);  v[192:207] packed P (kb, t) at 192 + 8 kb + 4 t
//   v[208:255] twelve fragment quads: Q rows (ring of 4: quads 0..3) | (dO row, V row kb 0, V row kb 1) triples (ring of 3: quads 4-6, 7-9, 10-11-0) |
//   transposed operands (ring of 4: quads 1, 2, 3, 4)
// ================================================================================================
// element order of the softmax riders: first the 16 elements the FIRST k-step of the second products needs (registers 0..7 of both key blocks),
// then the other 16: flat index i -> (key block, register)
static constexpr __device__ __host__ int b64_ekb(int i) { return (i >> 3) & 1; }
static constexpr __device__ __host__ int b64_ereg(int i) { return (i & 7) + 8 * (i >> 4); }

template <bool CAUSAL, bool ROPE, int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(64))) void attn_bwd_dkdv64w_kernel(AttnParams p) {
  static_assert(D == 128 || D == 96, "D");
  constexpr int NKS = D / 16, NOB = D / 32, NOPS = 2 * NOB;
  constexpr int A_DK = 0, A_DV = 128, V_KF = 64, V_S = 128, V_DP = 160, V_PKP = 192, V_FQ = 208;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  asm volatile("" ::: "v255", "a255");                   // the wave owns its SIMD's whole register file (see gemm_nt_256w4)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kl = lane & 31, hh = lane >> 5;
  const int hk = blockIdx.x, b = VP_BY(p);             // z (slowest dispatch index) = key block: early keys (most queries) first
  const int k0 = VP_BZ(p) * 256, kw0 = k0 + wave * 64;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int off = p.Skv - p.Sq, Sq = p.Sq, window = p.window;
  const int rep = p.Hq / p.Hkv;
  const float c = p.scale * LOG2E;
  const long nrows = (long)p.B * p.Hq * p.Sq;

  // ---- accumulators = 0 (the K fragments are loaded last, right before the ring starts: from then on v64..v255 are asm-owned)
  vp_static_for<2 * NOB * 16>([&](auto i_) __attribute__((always_inline)) {
    constexpr int i = decltype(i_)::value;
    R_AZERO(A_DK + 64 * (i / (NOB * 16)) + (i % (NOB * 16)));
    R_AZERO(A_DV + 64 * (i / (NOB * 16)) + (i % (NOB * 16)));
  });

  int qstart = 0, qend = Sq;
  if (CAUSAL) qstart = max(0, k0 - off) & ~31;
  if (window > 0) qend = min(Sq, k0 + 256 - off + window);
  if (k0 >= kvlen) qend = qstart;                      // whole key block is padding: gradients are zero
  const int ntq = qend > qstart ? (qend - qstart + 31) / 32 : 0;
  const int nit = (B64_ABL & 8) ? 0 : ntq * rep;       // flattened (head, q tile) iteration space  (ablation 8: prologue + epilogue only)

  // ---- LDS-DMA descriptors: Q / dO of this batch element (every head, rows past Sq read as zeros), the two statistics planes, V of this kv head
  const fwdm_u32x4s rsQ = attn_make_rs(p.q + (long)b * p.q_bs, (((long)Sq - 1) * p.q_ts + (long)p.Hq * D) * 2);
  const fwdm_u32x4s rsG = attn_make_rs(p.dout + (long)b * p.do_bs, (((long)Sq - 1) * p.do_ts + (long)p.Hq * D) * 2);
  const fwdm_u32x4s rsP = attn_make_rs(p.delta + (long)b * p.Hq * Sq, (nrows + (long)p.Hq * Sq) * 4);
  const fwdm_u32x4s rsV = attn_make_rs(p.v + (long)b * p.v_bs + (long)hk * D, (((long)p.Skv - 1) * p.v_ts + D) * 2);
  const uint32_t ldsb = attn_lds_addr(attn_smem);
  const uint32_t qts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.q_ts * 2)), gts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.do_ts * 2));
  const uint32_t vts2 = __builtin_amdgcn_readfirstlane((uint32_t)(p.v_ts * 2));
  uint32_t vQ, vG, vP;
  {
    const int da = lane >> 4, dch = (lane & 15) ^ (da << 2);
    vQ = (uint32_t)da * qts2 + (uint32_t)dch * 16u;
    vG = (uint32_t)da * gts2 + (uint32_t)dch * 16u;
    vP = lane < 32 ? (uint32_t)lane * 4u : (uint32_t)(nrows * 4) + (uint32_t)(lane - 32) * 4u;
    asm volatile("" : "+v"(vQ), "+v"(vG), "+v"(vP));
    // the wave's own 64 V rows -> LDS (16 pieces), once
    const uint32_t vV = (uint32_t)da * vts2 + (uint32_t)dch * 16u;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      ATTN_DMA16(ldsb + (uint32_t)(B64_KV_RING + (16 * wave + i) * B64_PIECE), vV, rsV, (uint32_t)(kw0 + 4 * i) * vts2);
  }
  const uint32_t m0w = __builtin_amdgcn_readfirstlane(ldsb + (uint32_t)wave * (uint32_t)B64_PIECE);
  // The tile stream (head within the GQA group, q tile) is walked by scalar byte offsets that advance by one tile per issue: Q / dO offset (+ 32 rows;
  // next head: back to the first row, + D features), statistics offset (+ 32 floats; next head: + Sq - 32 (ntq - 1)).  Past the end: the last tile again.
  uint32_t soQ = ((uint32_t)qstart + 4u * (uint32_t)wave) * qts2 + (uint32_t)(hk * rep) * (uint32_t)(D * 2);
  uint32_t soG = ((uint32_t)qstart + 4u * (uint32_t)wave) * gts2 + (uint32_t)(hk * rep) * (uint32_t)(D * 2);
  uint32_t soP = ((uint32_t)(hk * rep) * (uint32_t)Sq + (uint32_t)qstart) * 4u;
  int iq = 0, nissued = 0;
  const uint32_t backQ = (uint32_t)(D * 2) - (uint32_t)(ntq - 1) * 32u * qts2, backG = (uint32_t)(D * 2) - (uint32_t)(ntq - 1) * 32u * gts2;
  const uint32_t backP = ((uint32_t)Sq - (uint32_t)(ntq - 1) * 32u) * 4u;
  const uint32_t stP0 = __builtin_amdgcn_readfirstlane(ldsb + (uint32_t)(2 * B64_TILE));
  // piece PC (0 / 1: Q token groups wave, wave + 4; 2 / 3: dO likewise; 4: the statistics, wave 0 only) of the NEXT tile into stage ST (literal)
#define KV_PIECE(ST, PC)                                                                                        \
  {                                                                                                             \
    if constexpr ((PC) == 0) B64_DMA16(m0w + (uint32_t)((ST) * B64_KV_STAGE), vQ, rsQ, soQ);                    \
    if constexpr ((PC) == 1) B64_DMA16(m0w + (uint32_t)((ST) * B64_KV_STAGE + 4 * B64_PIECE), vQ, rsQ, soQ + 16u * qts2); \
    if constexpr ((PC) == 2) B64_DMA16(m0w + (uint32_t)((ST) * B64_KV_STAGE + B64_TILE), vG, rsG, soG);         \
    if constexpr ((PC) == 3) B64_DMA16(m0w + (uint32_t)((ST) * B64_KV_STAGE + B64_TILE + 4 * B64_PIECE), vG, rsG, soG + 16u * gts2); \
    if constexpr ((PC) == 4) { if (wave == 0) B64_DMA4(stP0 + (uint32_t)((ST) * B64_KV_STAGE), vP, rsP, soP); } \
  }
#define KV_ADVANCE()             /* (selects, no branches: this sits in the tile loop's issue stream) */         \
  {                                                                                                             \
    const bool adv_ = nissued + 1 < nit, wrap_ = iq + 1 == ntq;                                                 \
    nissued = min(nissued + 1, nit);                                                                            \
    soQ += adv_ ? (wrap_ ? backQ : 32u * qts2) : 0u;                                                            \
    soG += adv_ ? (wrap_ ? backG : 32u * gts2) : 0u;                                                            \
    soP += adv_ ? (wrap_ ? backP : 128u) : 0u;                                                                  \
    iq = adv_ ? (wrap_ ? 0 : iq + 1) : iq;                                                                      \
  }
#define KV_ISSUE(ST) { KV_PIECE(ST, 0) KV_PIECE(ST, 1) KV_PIECE(ST, 2) KV_PIECE(ST, 3) KV_PIECE(ST, 4) KV_ADVANCE() }

  // ---- fragment addresses (loop-invariant)
  uint32_t rowa[4], vrow[4], tra[NOB], sta;
  {
    const int rj = kl >> 2, ra = kl & 3;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      rowa[m] = ldsb + (uint32_t)(rj * B64_PIECE + ra * 256 + 16 * hh + 64 * (m ^ ra));
      vrow[m] = rowa[m] + (uint32_t)(B64_KV_RING + 16 * wave * B64_PIECE);
      asm volatile("" : "+v"(rowa[m]), "+v"(vrow[m]));
    }
    const int fr = lane & 15, ta = fr >> 2, tx = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), th = lane & 1;
#pragma unroll
    for (int db = 0; db < NOB; ++db) {
      tra[db] = ldsb + (uint32_t)(hh * B64_PIECE + ta * 256 + 64 * (db ^ ta) + 16 * tx + 8 * th);
      asm volatile("" : "+v"(tra[db]));
    }
    sta = ldsb + (uint32_t)(2 * B64_TILE + 16 * hh);                 // statistics: 4 floats at query 8 j + 4 hh  (+ 32 j bytes; delta plane + 128)
    asm volatile("" : "+v"(sta));
  }
  float cc = c, ninf = -INFINITY;
  asm volatile("" : "+v"(cc), "+v"(ninf));

  if constexpr (ROPE) asm volatile("" ::"s"(p.rope_cos), "s"(p.rope_sin), "s"(p.rope_pos));      // (the epilogue's kernel arguments: loaded here, not inside the stream)
  // ---- K fragments (B operands of S = Q K^T: lane = key, features 16 ks + 8 hh .. + 7) into v[64:127]: buffer loads with literal destinations
  B64_FENCE();
  {
    const fwdm_u32x4s rsKh = attn_make_rs(p.k + (long)b * p.k_bs + (long)hk * D, (((long)p.Skv - 1) * p.k_ts + D) * 2);
    uint32_t vok[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)                      // clamped; keys >= kv_len are masked (P = 0), keys >= Skv are not stored
      vok[kb] = (uint32_t)min(kw0 + 32 * kb + kl, p.Skv - 1) * (uint32_t)(p.k_ts * 2) + 16u * (uint32_t)hh;
    asm volatile("s_nop 4" ::: "memory");
    vp_static_for<2 * NKS>([&](auto i_) __attribute__((always_inline)) {
      constexpr int kb = decltype(i_)::value / NKS, ks = decltype(i_)::value % NKS;
      b64_use(vok, rsKh);
      asm volatile("buffer_load_dwordx4 v[%c2:%c3], %0, %1, 0 offen offset:%c4" ::"v"(vok[kb]), "s"(rsKh), "n"(V_KF + 32 * kb + 4 * ks),
                   "n"(V_KF + 32 * kb + 4 * ks + 3), "n"(32 * ks) : "memory");
    });
  }
  if (nit > 0) { KV_ISSUE(0) KV_ISSUE(1) KV_ISSUE(2) }
  B64_LGKM(0);                                          // (no scalar load of the prologue may be in flight inside the counted-lgkmcnt stream)
  int cq = 0;                                          // q tile (within its head) of the tile being consumed

  // ---- one tile.  ST = ring stage (literal).  The LDS reads of the whole kernel are ONE in-order stream with counted waits that runs on from tile to
  // tile: a tile ends by reading the NEXT tile's statistics (into the score blocks, free by then) and its first two Q rows.
  //   entry   (in flight: 4 statistics reads, Q rows 0, 1)
  //   phase 1 S' += Q K^T         reads: Q rows 2.., then the delta statistics into the dP' blocks + (dO row, V row, V row) triples 0, 1
  //   [mask]  compare / select statements on the score blocks (diagonal / ragged / window-edge tiles only)
  //   phase 2 dP' += dO V^T       reads: triples 2.., then transposed operands 0, 1, 2     riders: c S', exponentials 0..15, P packs of k-step 0
  //   barrier tile i + 1 landed for everybody, stage (i - 1) & 3 free: the LDS-DMA pieces of tile i + 3 go out under phase 3
  //   phase 3 dV^T += dO^T P      reads: operands 3..                                       riders: exponentials 16..31, P packs of k-step 1, dS, dS packs
  //   phase 4 dK^T += Q^T dS      reads: operands .., then the next tile's statistics and Q rows 0, 1
  // Fragment quads (v[208 + 4 i]): Q rows ring of 3 = quads 0, 1, 2 | triples: slot (ks + SH) % 3 = quads 3-5, 6-8, 9-11 | transposed operands ring of
  // 4 = quads 9, 10, 11, 3 (triple slot 2 is free when the first three are issued, slot 0 when the fourth is).
  constexpr int SH = NKS == 6 ? 2 : 0;
#define KV_FQ(I) (V_FQ + 4 * (I))
#define KV_TSLOT(KS, J) KV_FQ(3 + 3 * (((KS) + SH) % 3) + (J))
#define KV_QRD(KS, ST) R_RD128(KV_FQ((KS) % 3), rowa[(KS) >> 1], (ST) * B64_KV_STAGE + 32 * ((KS) & 1))
#define KV_GRD(KS, ST) R_RD128(KV_TSLOT(KS, 0), rowa[(KS) >> 1], (ST) * B64_KV_STAGE + B64_TILE + 32 * ((KS) & 1))
#define KV_VRD(KB, KS) R_RD128(KV_TSLOT(KS, 1 + (KB)), vrow[(KS) >> 1], (KB) * 8 * B64_PIECE + 32 * ((KS) & 1))
#define KV_TRIPLE(KS, ST) { KV_GRD(KS, ST); KV_VRD(0, KS); KV_VRD(1, KS); }
  // transposing-read stream of phases 3 + 4: operand u = 0 .. 2 NOPS - 1; u < NOPS: dO^T (t = u / NOB, db = u % NOB), else Q^T
#define KV_TRQ(U) KV_FQ(((U) & 3) == 3 ? 3 : 9 + ((U) & 3))
#define KV_TROFF(U, ST) ((ST) * B64_KV_STAGE + ((U) < NOPS ? B64_TILE : 0) + (((U) % NOPS) / NOB) * 4 * B64_PIECE)
#define KV_TRLO(U, ST) R_RDTR(KV_TRQ(U), tra[((U) % NOPS) % NOB], KV_TROFF(U, ST))
#define KV_TRHI(U, ST) R_RDTR(KV_TRQ(U) + 2, tra[((U) % NOPS) % NOB], KV_TROFF(U, ST) + 2 * B64_PIECE)
  // element G of the transposing stream (0 = op0.lo, 1 = op0.hi, 2 = op1.lo, ...)
#define KV_TRSTREAM(G, ST) { if constexpr (((G) & 1) == 0) { KV_TRLO((G) / 2, ST); } else { KV_TRHI((G) / 2, ST); } }
  // statistics of the tile in stage ST: J = 0..3 -> the lse plane into S'(kb = 0), registers 4 J ..; J = 4..7 -> the delta plane into dP'(kb = 0).  The
  // kb = 1 chains START from the kb = 0 block through their first MFMA's C operand (issued before kb = 0's first MFMA accumulates on it): 8 broadcast
  // reads per tile instead of 16
#define KV_STAT(J, ST) R_RD128(((J) < 4 ? V_S : V_DP) + 4 * ((J) & 3), sta, (ST) * B64_KV_STAGE + ((J) < 4 ? 0 : 128) + 32 * ((J) & 3))
  // cold start of the stream (first computed tile, or the tile behind one the wave skipped)
#define KV_COLD(ST) { vp_static_for<4>([&](auto j_) __attribute__((always_inline)) { b64_use(sta); KV_STAT(decltype(j_)::value, ST); }); KV_QRD(0, ST); KV_QRD(1, ST); }
  // rider slots (MFMA index within the phase; riders of an MFMA are issued right behind it)
  constexpr int MULPM = NKS == 8 ? 3 : 4;               // phase 2: c S' multiplies per MFMA from MFMA 2 on
  constexpr int EPM = NOB == 4 ? 3 : 4;                 // phase 3: exponentials 16..31 per MFMA from MFMA 0 on
  static_assert(2 + 31 / MULPM < 2 * NKS && 2 + 15 / MULPM + 4 < 2 * NKS, "phase 2 rider slots");
  static_assert((31 - 16) / EPM + 1 < 2 * NOB - 1, "P packs of the second k-step must be two states ahead of its first MFMA");
#define KV_TILE(ST)                                                                                             \
  {                                                                                                             \
    /* ---- phase 1 */                                                                                          \
    vp_static_for<NKS>([&](auto ks_) __attribute__((always_inline)) {                                           \
      constexpr int ks = decltype(ks_)::value;                                                                  \
      b64_use(rowa, vrow, sta);                                                                                 \
      if constexpr (ks + 2 < NKS) { KV_QRD(ks + 2, ST); B64_LGKM(2); }                                          \
      else if constexpr (ks + 2 == NKS) {                                                                       \
        vp_static_for<4>([&](auto j_) __attribute__((always_inline)) { b64_use(sta); KV_STAT(4 + decltype(j_)::value, ST); }); \
        KV_TRIPLE(0, ST)                                                                                        \
        B64_LGKM(8);                                                                                            \
      } else { KV_TRIPLE(1, ST) B64_LGKM(10); }                                                                 \
      if constexpr (ks == 0) { R_MF_C("v", "v", "v", V_S + 16, KV_FQ(0), V_KF + 32, V_S); }                     \
      else { R_MF("v", "v", "v", V_S + 16, KV_FQ(ks % 3), V_KF + 32 + 4 * ks); }                                \
      R_MF("v", "v", "v", V_S, KV_FQ(ks % 3), V_KF + 4 * ks);                                                   \
    });                                                                                                         \
    if (need_mask) {                                                                                            \
      asm volatile("s_nop 15" ::: "memory");        /* S' is 12 issue states old before the selects read it */   \
      int lm_ = threadIdx.x & 63;    /* (lane-derived values from an opaque lane id: nothing lane-dependent stays live across the loop) */ \
      asm volatile("" : "+v"(lm_));                                                                             \
      const int klm_ = lm_ & 31, hhm_ = lm_ >> 5;                                                               \
      const int eSq = Sq - q0 - 4 * hhm_;                                                                       \
      vp_static_for<2>([&](auto kb_) __attribute__((always_inline)) {                                           \
        constexpr int kb = decltype(kb_)::value;                                                                \
        b64_use(ninf);                                                                                          \
        const int key = kw0 + 32 * kb + klm_;                                                                   \
        const int eLo = key >= kvlen ? (1 << 30) : (CAUSAL ? key - off - q0 - 4 * hhm_ : -(1 << 30));           \
        const int eHi = window > 0 ? min(eSq, key - off + window - q0 - 4 * hhm_) : eSq;                        \
        vp_static_for<16>([&](auto r_) __attribute__((always_inline)) {                                         \
          constexpr int r = decltype(r_)::value;                                                                \
          b64_use(eLo, eHi, ninf);                                                                              \
          R_MASK(V_S + 16 * kb + r, (r & 3) + 8 * (r >> 2), eLo, eHi, ninf);                                    \
        });                                                                                                     \
      });                                                                                                       \
    }                                                                                                           \
    /* ---- phase 2 */                                                                                          \
    vp_static_for<NKS>([&](auto ks_) __attribute__((always_inline)) {                                           \
      constexpr int ks = decltype(ks_)::value;                                                                  \
      b64_use(rowa, vrow, tra, cc);                                                                             \
      if constexpr (ks + 2 < NKS) { KV_TRIPLE(ks + 2, ST) }                                                     \
      else { KV_TRSTREAM(3 * (ks + 2 - NKS), ST) KV_TRSTREAM(3 * (ks + 2 - NKS) + 1, ST) KV_TRSTREAM(3 * (ks + 2 - NKS) + 2, ST) } \
      B64_LGKM(6);                                                                                              \
      vp_static_for<2>([&](auto kb_) __attribute__((always_inline)) {                                           \
        constexpr int kb = 1 - decltype(kb_)::value;     /* kb = 1 first: at ks = 0 it reads the delta block of kb = 0 as C */ \
        constexpr int m = 2 * ks + decltype(kb_)::value;                                                        \
        b64_use(cc);                                                                                            \
        if constexpr (ks == 0 && kb == 1) { R_MF_C("v", "v", "v", V_DP + 16, KV_TSLOT(0, 0), KV_TSLOT(0, 2), V_DP); } \
        else { R_MF("v", "v", "v", V_DP + 16 * kb, KV_TSLOT(ks, 0), KV_TSLOT(ks, 1 + kb)); }                    \
        /* riders: multiply i under MFMA 2 + i / MULPM; exponential j (first 16 elements) two MFMAs behind its multiply; pack q two behind */ \
        vp_static_for<MULPM>([&](auto j_) __attribute__((always_inline)) {                                      \
          b64_use(cc);                                                                                          \
          constexpr int i = MULPM * (m - 2) + decltype(j_)::value;                                              \
          if constexpr (m >= 2 && i < 32) R_MULV(V_S + 16 * b64_ekb(i) + b64_ereg(i), cc);                      \
        });                                                                                                     \
        vp_static_for<16>([&](auto j_) __attribute__((always_inline)) {                                         \
          constexpr int j = decltype(j_)::value;                                                                \
          if constexpr (2 + j / MULPM + 2 == m) R_EXP(V_S + 16 * b64_ekb(j) + b64_ereg(j));                     \
        });                                                                                                     \
        vp_static_for<8>([&](auto q_) __attribute__((always_inline)) {                                          \
          constexpr int q = decltype(q_)::value;                                                                \
          if constexpr (2 + (2 * q + 1) / MULPM + 4 == m)                                                       \
            R_CVT(V_PKP + 8 * b64_ekb(2 * q) + (q & 3), V_S + 16 * b64_ekb(2 * q) + b64_ereg(2 * q), V_S + 16 * b64_ekb(2 * q) + b64_ereg(2 * q + 1)); \
        });                                                                                                     \
      });                                                                                                       \
    });                                                                                                         \
    /* ---- tile IT + 1 landed for everybody; stage (IT - 1) & 3 is free */                                     \
    if (wave == 0) { B64_VMCNT(5); } else { B64_VMCNT(4); }                                                     \
    B64_BAR();                                                                                                  \
    /* ---- phase 3 (+ the LDS-DMA pieces of tile IT + 3 into stage (ST + 3) & 3, one per operand step) */      \
    vp_static_for<NOPS>([&](auto n_) __attribute__((always_inline)) {                                           \
      constexpr int n = decltype(n_)::value;                                                                    \
      b64_use(tra, m0w, vQ, vG, vP, rsQ, rsG, rsP, soQ, soG, soP, qts2, gts2, stP0, wave);                      \
      KV_TRSTREAM(2 * n + 6, ST) KV_TRSTREAM(2 * n + 7, ST)                                                     \
      B64_LGKM(6);                                                                                              \
      vp_static_for<2>([&](auto kb_) __attribute__((always_inline)) {                                           \
        constexpr int kb = decltype(kb_)::value;                                                                \
        constexpr int m = 2 * n + kb;                                                                           \
        R_MF("a", "v", "v", A_DV + 64 * kb + 16 * (n % NOB), KV_TRQ(n), V_PKP + 8 * kb + 4 * (n / NOB));        \
        if constexpr (kb == 0 && n < 5) { b64_use(m0w, vQ, vG, vP, rsQ, rsG, rsP, soQ, soG, soP, qts2, gts2, stP0, wave); KV_PIECE(((ST) + 3) & 3, n) } \
        vp_static_for<EPM>([&](auto j_) __attribute__((always_inline)) {                                        \
          constexpr int j = 16 + EPM * m + decltype(j_)::value;                                                 \
          if constexpr (j < 32) R_EXP(V_S + 16 * b64_ekb(j) + b64_ereg(j));                                     \
        });                                                                                                     \
        vp_static_for<8>([&](auto q_) __attribute__((always_inline)) {                                          \
          constexpr int q = 8 + decltype(q_)::value;                                                            \
          if constexpr ((2 * q + 1 - 16) / EPM + 1 == m)                                                        \
            R_CVT(V_PKP + 8 * b64_ekb(2 * q) + 4 + (q & 3), V_S + 16 * b64_ekb(2 * q) + b64_ereg(2 * q), V_S + 16 * b64_ekb(2 * q) + b64_ereg(2 * q + 1)); \
        });                                                                                                     \
        vp_static_for<4>([&](auto j_) __attribute__((always_inline)) {                                          \
          constexpr int i = 4 * (m - 4) + decltype(j_)::value;                                                  \
          if constexpr (m >= 4 && i < 32) R_MULR(V_DP + 16 * b64_ekb(i) + b64_ereg(i), V_S + 16 * b64_ekb(i) + b64_ereg(i)); \
        });                                                                                                     \
        vp_static_for<16>([&](auto q_) __attribute__((always_inline)) {                                         \
          constexpr int q = decltype(q_)::value;                                                                \
          if constexpr (4 + (2 * q + 1) / 4 + 1 == m)                                                           \
            R_CVT(V_DP + 16 * b64_ekb(2 * q) + 8 * (q >> 3) + (q & 3), V_DP + 16 * b64_ekb(2 * q) + b64_ereg(2 * q), V_DP + 16 * b64_ekb(2 * q) + b64_ereg(2 * q + 1)); \
        });                                                                                                     \
      });                                                                                                       \
    });                                                                                                         \
    KV_ADVANCE()                                                                                                \
    /* ---- phase 4 (the dS packs that did not fit under phase 3 ride under its first MFMAs; the stream runs on into the next tile) */ \
    vp_static_for<NOPS>([&](auto n_) __attribute__((always_inline)) {                                           \
      constexpr int n = decltype(n_)::value;                                                                    \
      b64_use(tra, rowa, sta);                                                                                  \
      if constexpr (n + 3 < NOPS) { KV_TRSTREAM(2 * (NOPS + n) + 6, ST) KV_TRSTREAM(2 * (NOPS + n) + 7, ST) B64_LGKM(6); } \
      else if constexpr (n + 3 == NOPS) { B64_LGKM(4); }                                                        \
      else if constexpr (n + 3 == NOPS + 1) {                                                                   \
        vp_static_for<4>([&](auto j_) __attribute__((always_inline)) { b64_use(sta); KV_STAT(decltype(j_)::value, ((ST) + 1) & 3); }); \
        B64_LGKM(6);                                                                                            \
      } else { KV_QRD(0, ((ST) + 1) & 3); KV_QRD(1, ((ST) + 1) & 3); B64_LGKM(6); }                              \
      vp_static_for<2>([&](auto kb_) __attribute__((always_inline)) {                                           \
        constexpr int kb = decltype(kb_)::value;                                                                \
        constexpr int m = 2 * NOPS + 2 * n + kb;         /* continues phase 3's MFMA count */                    \
        R_MF("a", "v", "v", A_DK + 64 * kb + 16 * (n % NOB), KV_TRQ(NOPS + n), V_DP + 16 * kb + 8 * (n / NOB));  \
        vp_static_for<16>([&](auto q_) __attribute__((always_inline)) {                                         \
          constexpr int q = decltype(q_)::value;                                                                \
          if constexpr (4 + (2 * q + 1) / 4 + 1 == m)                                                           \
            R_CVT(V_DP + 16 * b64_ekb(2 * q) + 8 * (q >> 3) + (q & 3), V_DP + 16 * b64_ekb(2 * q) + b64_ereg(2 * q), V_DP + 16 * b64_ekb(2 * q) + b64_ereg(2 * q + 1)); \
        });                                                                                                     \
      });                                                                                                       \
    });                                                                                                         \
  }
  // One iteration: the wave computes the tile (cold start of the stream if the previous iteration skipped), or only keeps the block's cadence.
  // (the flag lives in an SGPR written by asm: as a C++ bool the compiler specialised every unrolled iteration on it — a flag machine of ~100 scalar
  // instructions and ~25 branches per iteration)
  int warm;
  asm volatile("s_mov_b32 %0, 0" : "=s"(warm));
#define KV_ITER(ST)                                                                                             \
  {                                                                                                             \
    B64_FENCE();                                                                                                \
    const int q0 = qstart + cq * 32;                                                                            \
    if (++cq == ntq) cq = 0;                                                                                    \
    /* wave-uniform skip: every query of this tile is below this wave's first key (causal), or every key of the wave is at or below the window's \
       lower bound of the tile's first query, or the wave's keys are all padding */                            \
    const bool active = (it + (ST) < nit) && (!CAUSAL || (q0 + 31 + off >= kw0)) && !(window > 0 && kw0 + 63 <= q0 + off - window) && kw0 < kvlen; \
    if (active) {                                                                                               \
      const bool need_mask = (q0 + 32 > Sq) || (kw0 + 64 > kvlen) || (CAUSAL && (kw0 + 63 > q0 + off)) ||       \
                             (window > 0 && kw0 <= q0 + 31 + off - window);                                     \
      if (!warm) KV_COLD(ST)                                                                                    \
      KV_TILE(ST)                                                                                               \
      asm volatile("s_mov_b32 %0, 1" : "=s"(warm));                                                             \
    } else {                                                                                                    \
      if (warm) B64_LGKM(0);                        /* (the previous tile's read-ahead of this one) */           \
      asm volatile("s_mov_b32 %0, 0" : "=s"(warm));                                                             \
      if (wave == 0) { B64_VMCNT(5); } else { B64_VMCNT(4); }                                                   \
      B64_BAR();                                                                                                \
      KV_ISSUE(((ST) + 3) & 3)                                                                                  \
    }                                                                                                           \
  }
  if (nit > 0) {
    if (wave == 0) { B64_VMCNT(10); } else { B64_VMCNT(8); }      // tile 0 landed (this wave's part; the V rows and K fragments are older)
    B64_BAR();
    // whole groups of four iterations (the ring stage is a literal): up to three trailing iterations past the last tile only keep the cadence (a
    // barrier and a re-fetch of the last tile each) — no exits in the middle of the unrolled body, which the compiler turned into a flag machine
    for (int it = 0; it < nit; it += 4) {
      KV_ITER(0)
      KV_ITER(1)
      KV_ITER(2)
      KV_ITER(3)
    }
    B64_LGKM(0);
  }
  B64_FENCE();
  // epilogue: the RoPE tables of both rows first (they fly while the trailing DMAs drain), then the accumulators (lane-derived values re-derived
  // from an opaque lane id: see the dQ kernel)
  B64_FENCE();
  int ln_ = threadIdx.x & 63;
  asm volatile("" : "+v"(ln_));
  const int kl_ = ln_ & 31, hh_ = ln_ >> 5;
  if constexpr (ROPE) {
    uint32_t vo[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int keyc = min(kw0 + 32 * kb + kl_, p.Skv - 1);
      const long pp = (p.rope_pos ? (long)p.rope_pos[(long)b * p.Skv + keyc] : (long)keyc) * (D / 2);
      vo[kb] = (uint32_t)(pp * 4 + 16 * hh_);
    }
    b64_rope_issue<NOB>(p.rope_cos, p.rope_sin, vo);
  }
  B64_VMCNT(0);                                         // the tables landed; trailing (dummy) DMAs landed too ...
  B64_BAR();                                            // ... for every wave, and every wave is behind its last fragment read: the ring is free
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the last MFMAs' results are visible to v_accvgpr_read
  {   // whole-line stores through a wave-private staging slice in the (finished) ring, one 32-key block of dK / dV at a time
    bf16_t* stg = (bf16_t*)attn_smem + wave * 4096;
    bf16_t* gk = p.dk + (long)b * p.dk_bs + (long)kw0 * p.dk_ts + (long)hk * D;
    bf16_t* gv = p.dv + (long)b * p.dv_bs + (long)kw0 * p.dv_ts + (long)hk * D;
    b64_store_row<ROPE, NOB, A_DK, 0, true>(stg, p.scale, hh_, kl_);
    b64_flush_rows<D>(stg, gk, p.dk_ts, p.Skv - kw0, ln_);
    b64_store_row<false, NOB, A_DV, 0, true>(stg, 1.f, hh_, kl_);
    b64_flush_rows<D>(stg, gv, p.dv_ts, p.Skv - kw0, ln_);
    b64_store_row<ROPE, NOB, A_DK + 64, 1, true>(stg, p.scale, hh_, kl_);
    b64_flush_rows<D>(stg, gk + 32 * (long)p.dk_ts, p.dk_ts, p.Skv - kw0 - 32, ln_);
    b64_store_row<false, NOB, A_DV + 64, 1, true>(stg, 1.f, hh_, kl_);
    b64_flush_rows<D>(stg, gv + 32 * (long)p.dv_ts, p.dv_ts, p.Skv - kw0 - 32, ln_);
  }
#undef KV_PIECE
#undef KV_ADVANCE
#undef KV_ISSUE
#undef KV_STAT
#undef KV_COLD
#undef KV_ITER
#undef KV_FQ
#undef KV_TSLOT
#undef KV_QRD
#undef KV_GRD
#undef KV_VRD
#undef KV_TRIPLE
#undef KV_TRQ
#undef KV_TROFF
#undef KV_TRLO
#undef KV_TRHI
#undef KV_TRSTREAM
#undef KV_TILE
}
