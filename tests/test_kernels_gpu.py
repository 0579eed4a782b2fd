"""GPU parity of every C-ABI kernel against a CPU fp32 restatement (oracle functions where the op is a
reference op, plain torch math otherwise).  Inputs are bf16-rounded on the host so both sides see the
same values; tolerances are stated per test (bf16 outputs: ~2^-8 relative)."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from visper_lm_amd import ops as o
    return o


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape) * 7919 + len(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def dev(t):
    return t.cuda()


def close(got, ref, rtol=2e-2, atol=None, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if atol is None:
        atol = 1e-2 * float(ref.abs().max()) + 1e-6
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {float(err.max()):.4g} (ref max {float(ref.abs().max()):.4g})"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,generic", [(256, 256, 256, False), (300, 200, 128, False), (128, 128, 64, False),
                                           (77, 50, 40, True), (300, 200, 128, True), (1, 1024, 64, False),
                                           (577 * 2, 1024, 1024, False)])
def test_gemm_plain(ops, M, N, K, generic):
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=0.1, seed=2)
    ref = a.float() @ w.float().t()
    out = ops.gemm(dev(a), dev(w), force_generic=generic)
    close(out, ref, what=f"gemm {M}x{N}x{K}")
    out32 = ops.gemm(dev(a), dev(w), out_f32=True, force_generic=generic)
    close(out32, ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()), what="gemm f32 out")


@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(ops, epi):
    M, N, K = 200, 328, 192
    a, w, b, r = rnd(M, K, seed=3), rnd(N, K, scale=0.1, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
    y = a.float() @ w.float().t() + b.float()
    y = y.to(BF).float()
    if epi == 1:
        y = F.gelu(y)
    elif epi == 2:
        y = y * torch.sigmoid(1.702 * y)
    elif epi == 3:
        y = F.relu(y)
    ref = y.to(BF).float() + r.float()
    out = ops.gemm(dev(a), dev(w), bias=dev(b), residual=dev(r), epi=epi)
    close(out, ref, what=f"gemm epi {epi}")


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (700, 300, 192), (256, 1024, 64)])
def test_gemm_256_tile_kernel(ops, M, N, K):
    a, w, b, r = rnd(M, K, seed=50), rnd(N, K, scale=0.1, seed=51), rnd(N, seed=52), rnd(M, N, seed=53)
    ref = (a.float() @ w.float().t() + b.float()).to(BF).float() + r.float()
    out = ops.gemm(dev(a), dev(w), bias=dev(b), residual=dev(r), force_generic=3)
    close(out, ref, what=f"gemm256 {M}x{N}x{K}")
    out2 = ops.gemm(dev(a), dev(w), bias=dev(b), residual=dev(r), force_generic=2)
    close(out2, ref, what=f"gemm128 {M}x{N}x{K}")
    o = ops.gemm(dev(a), dev(w), bias=dev(b), residual=dev(r), force_generic=7)      # 8-phase kernel forced on a small problem
    close(o, ref, what=f"gemm 8-phase {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (256, 768, 384), (4608, 4096, 384), (8192, 4096, 512), (1024, 1024, 4096),
                                   (8192, 6144, 1024), (2048, 4096, 14336), (16384, 4096, 256)])
def test_gemm_4wave_kernel(ops, M, N, K):
    """One-wave-per-SIMD 256-tile kernel (force code 8): plain, bias + GELU + residual epilogue, fp32 output; single tiles, a persistent grid with
    1-2 tiles per block and the super-block walk."""
    a, w, b, r = rnd(M, K, seed=70), rnd(N, K, scale=0.1, seed=71), rnd(N, seed=72), rnd(M, N, seed=73)
    ag, wg = dev(a), dev(w)
    ref = a.float() @ w.float().t()
    close(ops.gemm(ag, wg, force_generic=8), ref, what=f"w4 {M}x{N}x{K}")
    assert torch.equal(ops.gemm(ag, wg, force_generic=8), ops.gemm(ag, wg, force_generic=7))       # same k order per MFMA chain as the 8-phase kernel
    assert torch.equal(ops.gemm(ag, wg, force_generic=13), ops.gemm(ag, wg, force_generic=7))      # 4-phase variant of the 8-phase kernel
    o32 = ops.gemm(ag, wg, out_f32=True, force_generic=8)
    assert o32.dtype == torch.float32
    assert float((o32.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * max(1.0, K / 256) ** 0.5
    if M * N <= 1 << 20:
        ref2 = F.gelu((ref + b.float()).to(BF).float()).to(BF).float() + r.float()
        close(ops.gemm(ag, wg, bias=dev(b), residual=dev(r), epi=ops.EPI_GELU, force_generic=8), ref2, what="w4 epilogue")



def test_gemm_8phase_large_k_and_edges(ops):
    """default large-problem kernel: many K-tiles (piece pipeline wraps both buffers), ragged M/N edges, auto dispatch."""
    M, N, K = 4100, 3080, 1024
    a, w = rnd(M, K, seed=54), rnd(N, K, scale=0.05, seed=55)
    ref = a.float() @ w.float().t()
    close(ops.gemm(dev(a), dev(w)), ref, what="gemm auto (8-phase)")
    close(ops.gemm(dev(a), dev(w), force_generic=3), ref, what="gemm persistent 256")


def test_gemm_strided_views(ops):
    # A is a column slice of a wider buffer, C is written into a column slice (fused-QKV style)
    M, K, N = 192, 128, 64
    big = rnd(M, 3 * K, seed=7)
    w = rnd(N, K, scale=0.1, seed=8)
    a = big[:, K:2 * K]
    ref = a.float() @ w.float().t()
    bigd = dev(big)
    outbuf = torch.zeros(M, 3 * N, device="cuda", dtype=BF)
    ops.gemm(bigd[:, K:2 * K], dev(w), out=outbuf[:, N:2 * N])
    close(outbuf[:, N:2 * N], ref, what="strided gemm")
    assert float(outbuf[:, :N].abs().max()) == 0 and float(outbuf[:, 2 * N:].abs().max()) == 0


def test_transpose_batched(ops):
    x = rnd(5, 136, 200, seed=77)                                      # edge tiles in both directions
    got = ops.transpose_batched(dev(x))
    assert torch.equal(got.cpu(), x.transpose(1, 2).contiguous())
    y = rnd(8, 1536, 576, seed=78)                                     # the seg targets' shape
    assert torch.equal(ops.transpose_batched(dev(y)).cpu(), y.transpose(1, 2).contiguous())


def test_transpose(ops):
    x = rnd(130, 75, seed=9)
    out = ops.transpose(dev(x))
    assert torch.equal(out.cpu(), x.t().contiguous())


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,H", [(37, 64), (19, 1024), (9, 4096), (5, 1536)])
def test_rmsnorm(ops, M, H):
    from oracle import visper_oracle as O
    x, w, dy, dres = rnd(M, H, seed=10), (1 + 0.1 * rnd(H, seed=11).float()).to(BF), rnd(M, H, seed=12), rnd(M, H, seed=13)
    xr = x.float().requires_grad_(True)
    y_ref = O.rms_norm(xr, w.float(), 1e-5)
    y_ref.backward(dy.float())
    y, rstd = ops.rmsnorm_fwd(dev(x), dev(w), 1e-5)
    close(y, y_ref, what="rmsnorm fwd")
    dx = ops.rmsnorm_bwd(dev(dy), dev(x), dev(w), rstd, dres=dev(dres))
    close(dx, xr.grad + dres.float(), what="rmsnorm bwd")


@pytest.mark.parametrize("M,H", [(37, 64), (19, 1024), (300, 1536)])
def test_layernorm(ops, M, H):
    x, w, b, dy = rnd(M, H, seed=14), (1 + 0.1 * rnd(H, seed=15).float()).to(BF), rnd(H, scale=0.1, seed=16), rnd(M, H, seed=17)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    y_ref = F.layer_norm(xr, (H,), wr, br, 1e-5)
    y_ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(w), dev(b), 1e-5)
    close(y, y_ref, what="layernorm fwd")
    dx, dw, db = ops.layernorm_bwd(dev(dy), dev(x), dev(w), mean, rstd)
    close(dx, xr.grad, what="layernorm dx")
    close(dw, wr.grad, rtol=1e-2, what="layernorm dw")
    close(db, br.grad, rtol=1e-2, what="layernorm db")


# ------------------------------------------------------------------------------------------------ elementwise
def test_rope_fwd_and_inverse(ops):
    from oracle import visper_oracle as O
    B, S, nh, hd = 2, 50, 3, 64
    x = rnd(B, S, nh * hd + 32, seed=18)                  # row-strided buffer, rope on the first nh*hd columns
    pos = torch.arange(S)[None].expand(B, S)
    cos, sin = O.rope_tables(pos, hd, 10000.0, BF)
    q = x[..., :nh * hd].reshape(B, S, nh, hd).transpose(1, 2).float()
    c, s = cos[:, None].float(), sin[:, None].float()
    ref = (q * c + O._rot_half(q) * s).transpose(1, 2).reshape(B, S, nh * hd)
    cs, sn = ops.rope_tables(S, hd, 10000.0, "cuda")
    xd = dev(x).reshape(B * S, -1)
    ops.rope_(xd, B * S, S, nh, hd, cs, sn)
    close(xd[:, :nh * hd].reshape(B, S, -1), ref, what="rope fwd")
    assert torch.equal(xd[:, nh * hd:].cpu(), x.reshape(B * S, -1)[:, nh * hd:])
    # inverse == autograd transpose of the rotation
    g = rnd(B, S, nh * hd, seed=19)
    qq = torch.randn(B, nh, S, hd, requires_grad=True)
    (qq * c + O._rot_half(qq) * s).backward(g.reshape(B, S, nh, hd).transpose(1, 2).float())
    gd = dev(g).reshape(B * S, -1).clone()
    ops.rope_(gd, B * S, S, nh, hd, cs, sn, inverse=True)
    close(gd.reshape(B, S, nh, hd), qq.grad.transpose(1, 2), what="rope inverse")


@pytest.mark.parametrize("B,S,nh,nkv,K", [(2, 256, 4, 2, 512), (3, 200, 6, 1, 256), (8, 2048, 32, 8, 128), (1, 300, 2, 2, 1024)])
def test_gemm_rope_epilogue(ops, B, S, nh, nkv, K):
    """vp_gemm_bf16_rope: the fused-QKV GEMM with RoPE of the q and k heads in its epilogue is bit-identical to vp_gemm_bf16 followed by vp_rope
    (v columns untouched), with and without explicit position ids, on M tails (M = B S not a multiple of 256) and with a row scale (= the plain
    GEMM of the row-scaled result: RMSNorm's 1/rms when gamma is folded into the weight)."""
    hd = 128
    M, N = B * S, (nh + 2 * nkv) * hd
    assert ops.gemm_rope_ok(M, N, K, hd) == (((M + 255) // 256) * (N // 256) >= 64)      # the engine's routing predicate; the C entry takes every case here
    a, w = dev(rnd(M, K, seed=300)), dev(rnd(N, K, scale=0.1, seed=301))
    cs, sn = ops.rope_tables(S, hd, 500000.0, "cuda")
    ref = ops.gemm(a, w, force_generic=7)
    ops.rope_(ref, M, S, nh + nkv, hd, cs, sn)
    got = ops.gemm_rope(a, w, S, (nh + nkv) * hd, cs, sn)
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    pos = torch.randint(0, S, (M,), dtype=torch.int32, device="cuda")
    ref2 = ops.gemm(a, w, force_generic=7)
    ops.rope_(ref2, M, S, nh + nkv, hd, cs, sn, pos=pos)
    assert torch.equal(ops.gemm_rope(a, w, S, (nh + nkv) * hd, cs, sn, pos=pos), ref2)
    # row scale: fp32 accumulator * scale -> bf16 (one rounding), then the rotation
    rs = (torch.rand(M, device="cuda") + 0.5).float()
    acc = ops.gemm(a, w, out_f32=True, force_generic=7)
    ref3 = (acc * rs[:, None]).to(BF)
    ops.rope_(ref3, M, S, nh + nkv, hd, cs, sn)
    got3 = ops.gemm_rope(a, w, S, (nh + nkv) * hd, cs, sn, row_scale=rs)
    d3 = (got3.float() - ref3.float()).abs()
    assert float((d3 > 0).float().mean()) < 1e-3 and float((d3 / ref3.float().abs().clamp_min(1e-2)).max()) < 1e-2      # (fp32 output path sums in another order)


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (4096, 4096, 256), (2048, 1024, 1024)])
def test_gemm_sumsq_and_rstd(ops, M, N, K):
    """vp_gemm_bf16_sumsq: the residual GEMM's output is bit-identical to vp_gemm_bf16's, and its per-16-column sums of squares finish
    (vp_rstd_from_sumsq) to the rstd vp_rmsnorm_fwd computes from the stored row (up to the last bf16 rounding of the row: the partials are
    taken in fp32 before it); two runs are bitwise identical."""
    a, w, r = dev(rnd(M, K, seed=310)), dev(rnd(N, K, scale=0.1, seed=311)), dev(rnd(M, N, seed=312))
    out, part = ops.gemm_sumsq(a, w, r)
    assert torch.equal(out, ops.gemm(a, w, residual=r, force_generic=7))
    assert part.shape == (M, N // 16)
    ref_part = out.float().view(M, N // 16, 16).pow(2).sum(-1)
    # layout: 16-column group j of the row = columns 128 (j // 8) + 64 ((j % 8) // 4) + 8 (j % 4) + {0..7} and + 32 + {0..7}
    cols = torch.arange(N, device="cuda").view(N // 128, 2, 2, 4, 8)               # [block, h, s2, g, e]
    grp = cols.permute(0, 1, 3, 2, 4).reshape(N // 16, 16)                          # [block, h, g] -> (s2, e)
    ref_part = out.float()[:, grp].pow(2).sum(-1)
    assert float(((part - ref_part).abs() / ref_part.clamp_min(1e-3)).max()) < 2e-2
    eps = 1e-5
    rstd = ops.rstd_from_sumsq(part, N, eps)
    _, rstd_ref = ops.rmsnorm_fwd(out, torch.ones(N, device="cuda", dtype=BF), eps)
    assert float(((rstd - rstd_ref).abs() / rstd_ref).max()) < 2e-3, float(((rstd - rstd_ref).abs() / rstd_ref).max())
    out2, part2 = ops.gemm_sumsq(a, w, r)
    assert torch.equal(part, part2) and torch.equal(rstd, ops.rstd_from_sumsq(part2, N, eps))


def test_gemm_swiglu_row_scale(ops):
    """Fused SwiGLU forward with an fp32 row scale on the accumulators (RMSNorm folded into the weight): gate_up = bf16(acc * scale), act from it."""
    M, N, K = 4096, 6144, 256
    x, wgu = dev(rnd(M, K, seed=320)), ops.interleave_gate_up(dev(rnd(N, K, seed=321) * 0.1))
    rs = (torch.rand(M, device="cuda") + 0.5).float()
    gu, act = ops.gemm_swiglu_fwd(x, wgu, row_scale=rs)
    acc = ops.gemm(x, wgu, out_f32=True, force_generic=7)
    gu_ref = (acc * rs[:, None]).to(BF)
    d = (gu.float() - gu_ref.float()).abs()
    assert float((d > 0).float().mean()) < 1e-3 and float((d / gu_ref.float().abs().clamp_min(1e-2)).max()) < 1e-2
    assert torch.equal(act, ops.swiglu_fwd(gu))
    gu1, act1 = ops.gemm_swiglu_fwd(x, wgu, row_scale=torch.ones(M, device="cuda"))
    gu0, act0 = ops.gemm_swiglu_fwd(x, wgu)
    assert torch.equal(gu1, gu0) and torch.equal(act1, act0)
    with pytest.raises(Exception):                             # not a one-wave-per-SIMD shape: refused, never silently unscaled
        ops.gemm_swiglu_fwd(x[:256], wgu[:256], row_scale=rs[:256])


def _ileave(t, Fd):
    """[gate | up] column halves -> the library's 8-wide chunk interleave (g0..7 | u0..7 | g8..15 | ...)."""
    g, u = t[..., :Fd], t[..., Fd:]
    return torch.stack([g.reshape(*g.shape[:-1], Fd // 8, 8), u.reshape(*u.shape[:-1], Fd // 8, 8)], -2).reshape(*t.shape)


def test_swiglu(ops):
    M, Fd = 33, 136
    gu, d = rnd(M, 2 * Fd, seed=20), rnd(M, Fd, seed=21)
    gr = gu.float().requires_grad_(True)
    ref = F.silu(gr[:, :Fd]) * gr[:, Fd:]
    ref.backward(d.float())
    gu_i = dev(_ileave(gu, Fd))
    close(ops.swiglu_fwd(gu_i), ref, what="swiglu fwd")
    close(ops.swiglu_bwd(dev(d), gu_i), _ileave(gr.grad, Fd), what="swiglu bwd")
    w = rnd(2 * Fd, 24, seed=24)
    assert torch.equal(ops.interleave_gate_up(dev(w)).cpu().t(), _ileave(w.t().contiguous(), Fd))


def test_gemm_swiglu_fused(ops):
    """Fused epilogues == GEMM followed by the standalone SwiGLU kernels, bit for bit (same rounding points)."""
    M, Fd, H = 512, 768, 256
    x, wgu, wd = rnd(M, H, seed=25), rnd(2 * Fd, H, seed=26) * 0.1, rnd(H, Fd, seed=27) * 0.1
    dy = rnd(M, H, seed=28)
    xg, dyg = dev(x), dev(dy)
    wgu_i = ops.interleave_gate_up(dev(wgu))
    wd_T = ops.transpose(dev(wd))                     # [Fd, H]
    gu_ref = ops.gemm(xg, wgu_i)
    act_ref = ops.swiglu_fwd(gu_ref)
    gu, act = ops.gemm_swiglu_fwd(xg, wgu_i)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    dgu_ref = ops.swiglu_bwd(ops.gemm(dyg, wd_T), gu_ref)
    dgu = ops.gemm_swiglu_bwd(dyg, wd_T, gu_ref)
    # same formula, but hipcc may contract the fp32 multiply-adds differently in the two kernels: allow isolated 1-ulp flips
    diff = (dgu.float() - dgu_ref.float()).abs()
    assert float((diff > 0).float().mean()) < 1e-4 and float((diff / dgu_ref.float().abs().clamp_min(1e-3)).max()) < 1e-2
    # and against fp32 torch
    gr = (x.float() @ wgu.float().t()).requires_grad_(True)
    ref = F.silu(gr[:, :Fd]) * gr[:, Fd:]
    close(act, ref, what="fused swiglu fwd")
    ref.backward(dy.float() @ wd.float())
    close(dgu, _ileave(gr.grad, Fd), what="fused swiglu bwd")


@pytest.mark.parametrize("M,N,K", [(4616, 3072, 1024), (577, 1024, 4096), (300, 256, 128), (4608, 1536, 384), (18432, 1536, 256), (2000, 4096, 256)])
def test_gemm_4wave_general_variant(ops, M, N, K):
    """General variant of the 4-wave kernel (force code 14): bias / activation / residual epilogues and an M tail (ViT's M = 8 x 577, a single
    partial row tile, tails that end inside the first / second wave row).  Bit-identical to the 8-phase kernel's epilogue (same rounding points:
    acc + bias -> bf16 -> activation -> bf16 -> + residual -> bf16); rows past M of a longer output buffer are not written."""
    a, w, b, r = rnd(M, K, seed=270), rnd(N, K, scale=0.1, seed=271), rnd(N, seed=272), rnd(M, N, seed=273)
    ag, wg, bg, rg = dev(a), dev(w), dev(b), dev(r)
    ref = a.float() @ w.float().t() + b.float()
    close(ops.gemm(ag, wg, bias=bg, force_generic=14), ref, what=f"w4g bias {M}x{N}x{K}")
    for kw in (dict(bias=bg), dict(bias=bg, residual=rg), dict(residual=rg), dict(), dict(bias=bg, epi=ops.EPI_GELU), dict(bias=bg, epi=ops.EPI_QUICK_GELU),
               dict(bias=bg, epi=ops.EPI_RELU), dict(epi=ops.EPI_RELU)):
        got, want = ops.gemm(ag, wg, force_generic=14, **kw), ops.gemm(ag, wg, force_generic=7, **kw)
        assert torch.equal(got, want), (M, N, K, sorted(kw), float((got.float() - want.float()).abs().max()))
    guard = torch.full((M + 300, N), 3.0, device="cuda", dtype=BF)
    ops.gemm(ag, wg, bias=bg, residual=rg, out=guard[:M], force_generic=14)
    assert torch.equal(guard[:M], ops.gemm(ag, wg, bias=bg, residual=rg, force_generic=7))
    assert float((guard[M:] - 3.0).abs().max()) == 0
    # the automatic choice takes it for these launches (and is the same result)
    assert torch.equal(ops.gemm(ag, wg, bias=bg, epi=ops.EPI_QUICK_GELU), ops.gemm(ag, wg, bias=bg, epi=ops.EPI_QUICK_GELU, force_generic=7))


@pytest.mark.parametrize("M,N,K", [(4096, 6144, 256), (2048, 12288, 384)])
def test_gemm_4wave_epilogues_whole_line_stores(ops, M, N, K):
    """The 4-wave kernel's epilogues store whole 128-byte lines (rows 8..15 of a 16-row block trade halves with rows 0..7 by DPP): residual
    epilogue bit-identical to the 8-phase kernel's, C and the residual as column slices of wider buffers whose base is only 16-byte aligned
    (lines then straddle), nothing written outside the slice; fused SwiGLU forward / backward at shapes that ROUTE to the 4-wave kernel
    (>= 192 tiles) against the GEMM + standalone SwiGLU kernels."""
    a, w, r = rnd(M, K, seed=170), rnd(N, K, scale=0.1, seed=171), rnd(M, N, seed=172)
    ag, wg, rg = dev(a), dev(w), dev(r)
    assert torch.equal(ops.gemm(ag, wg, residual=rg, force_generic=8), ops.gemm(ag, wg, residual=rg, force_generic=7))
    wide = torch.full((M, N + 512), 7.0, device="cuda", dtype=BF)
    rwide = torch.zeros(M, N + 264, device="cuda", dtype=BF)
    rwide[:, 8:8 + N] = rg
    ops.gemm(ag, wg, residual=rwide[:, 8:8 + N], out=wide[:, 8:8 + N], force_generic=8)
    assert torch.equal(wide[:, 8:8 + N], ops.gemm(ag, wg, residual=rg, force_generic=7))
    assert float((wide[:, :8] - 7.0).abs().max()) == 0 and float((wide[:, 8 + N:] - 7.0).abs().max()) == 0
    # fused SwiGLU on the 4-wave kernel: N = 2 F gate|up columns forward, F columns backward
    Fd = N // 2
    assert (M // 256) * (Fd // 256) >= 192                       # both directions route to gemm_nt_256w4
    x, wgu, dy, wd = rnd(M, K, seed=173), rnd(N, K, seed=174) * 0.1, rnd(M, K, seed=175), rnd(K, Fd, seed=176) * 0.1
    xg, dyg = dev(x), dev(dy)
    wgu_i = ops.interleave_gate_up(dev(wgu))
    wd_T = ops.transpose(dev(wd))                     # [Fd, K]
    gu_ref = ops.gemm(xg, wgu_i, force_generic=7)
    act_ref = ops.swiglu_fwd(gu_ref)
    gu, act = ops.gemm_swiglu_fwd(xg, wgu_i)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    dgu_ref = ops.swiglu_bwd(ops.gemm(dyg, wd_T, force_generic=7), gu_ref)
    dgu = ops.gemm_swiglu_bwd(dyg, wd_T, gu_ref)
    diff = (dgu.float() - dgu_ref.float()).abs()
    assert float((diff > 0).float().mean()) < 1e-4 and float((diff / dgu_ref.float().abs().clamp_min(1e-3)).max()) < 1e-2


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 192), (256, 1024, 4608), (4352, 4096, 256), (8192, 4096, 1024)])
def test_gemm_tn_weight_gradient(ops, M, N, K):
    """dW = dY^T X straight from row-major activations == fp32 matmul of the same bf16 operands (fp32 output, strided operands,
    accumulate mode, persistent (> 256 tiles) and one-block-per-tile grids, odd and even K-tile counts)."""
    dy, x = rnd(K, M + 8, seed=61), rnd(K, N, seed=62)
    dyg = dev(dy)[:, 8:]                                # 16-byte aligned column offset, ld > M
    xg = dev(x)
    ref = dy[:, 8:].float().t() @ x.float()
    out = ops.gemm_tn(dyg, xg)
    assert out.dtype == torch.float32
    scale = float(ref.abs().max())
    assert float((out.cpu() - ref).abs().max()) <= 2e-5 * scale * max(1.0, K / 256) ** 0.5
    ops.gemm_tn(dyg, xg, out=out, accumulate=True)
    assert float((out.cpu() - 2 * ref).abs().max()) <= 4e-5 * scale * max(1.0, K / 256) ** 0.5
    if M * N <= 512 * 1024:
        ob = ops.gemm_tn(dyg, xg, out_f32=False)
        close(ob, ref, what="tn bf16 out")
    # same numbers as the path it replaces (transposes + NT GEMM), up to fp32 summation order inside one MFMA
    old = ops.gemm(ops.transpose(dyg.contiguous()), ops.transpose(xg), out_f32=True)
    ops.gemm_tn(dyg, xg, out=out)
    assert float((out - old).abs().max()) <= 1e-5 * scale * max(1.0, K / 256) ** 0.5
    with pytest.raises(RuntimeError):
        ops._lib.call("vp_gemm_tn_bf16", 200, 256, 64, dyg.data_ptr(), dyg.stride(0), xg.data_ptr(), N, out.data_ptr(), N, 1, 0, None, None)


@pytest.mark.parametrize("D,theta", [(128, 500000.0), (96, 10000.0)])
def test_attn_bwd_fused_rope(ops, D, theta):
    """dq / dk rotated back inside the D = 128 (Llama) and D = 96 (Phi-3) backward kernels == vp_attn_bwd followed by vp_rope(inverse), bit for
    bit (GQA, ragged S, explicit position ids)."""
    B, Hq, Hkv, S = 2, 8, 2, 300
    qkv = dev(rnd(B, S, (Hq + 2 * Hkv) * D, seed=80))
    q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D))
    v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
    do = dev(rnd(B, S, Hq, D, seed=81))
    o, lse = ops.attn_fwd(q, k, v, True)
    cos_t, sin_t = ops.rope_tables(S, D, theta, "cuda")
    d1 = torch.zeros_like(qkv); d2 = torch.zeros_like(qkv)
    views = lambda d: (d[..., :Hq * D].unflatten(-1, (Hq, D)), d[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)),
                       d[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D)))
    a, b_, c = views(d1)
    ops.attn_bwd(q, k, v, o, lse, do, True, dq=a, dk=b_, dv=c)
    ops.rope_(d1.view(B * S, -1), B * S, S, Hq + Hkv, D, cos_t, sin_t, inverse=True)
    a, b_, c = views(d2)
    ops.attn_bwd(q, k, v, o, lse, do, True, dq=a, dk=b_, dv=c, rope=(cos_t, sin_t))
    assert torch.equal(d1, d2)
    pos = (torch.arange(S, device="cuda", dtype=torch.int32)[None, :] + torch.tensor([[3], [0]], device="cuda", dtype=torch.int32)).clamp_max(S - 1).contiguous()
    d3 = torch.zeros_like(qkv)
    a, b_, c = views(d1)
    ops.attn_bwd(q, k, v, o, lse, do, True, dq=a, dk=b_, dv=c)
    ops.rope_(d1.view(B * S, -1), B * S, S, Hq + Hkv, D, cos_t, sin_t, pos=pos.view(-1), inverse=True)
    a, b_, c = views(d3)
    ops.attn_bwd(q, k, v, o, lse, do, True, dq=a, dk=b_, dv=c, rope=(cos_t, sin_t, pos))
    assert torch.equal(d1, d3)


def test_gemm_dynamic_tile_scheduling(ops):
    """Per-XCD dynamic tile claims == the static walk, bit for bit, launch after launch (the kernel re-arms its counters), with and without
    a stand-in collective holding 24 CUs; also the fused SwiGLU launches."""
    M, N, K = 4608, 4096, 256                         # 18 x 16 = 288 tiles: persistent grid, 1-2 tiles per block
    a, w = dev(rnd(M, K, seed=90)), dev(rnd(N, K, scale=0.1, seed=91))
    ref = ops.gemm(a, w)
    M2, F2 = 8192, 4096                               # 32 x 32 tiles of [gate|up]
    x2, wgu = dev(rnd(M2, 256, seed=92)), ops.interleave_gate_up(dev(rnd(2 * F2, 256, scale=0.1, seed=93)))
    gu_ref, act_ref = ops.gemm_swiglu_fwd(x2, wgu)
    dyt, xt = dev(rnd(512, 4352, seed=94)), dev(rnd(512, 4096, seed=95))       # TN weight gradient, 17 x 16 tiles
    tn_ref = ops.gemm_tn(dyt, xt)
    prev = ops.set_dynamic(True)               # counter blocks: caller-owned, one zeroed block per stream (ops._sched)
    try:
        for it in range(4):
            if it == 2:
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    with ops._lib.debug_library():             # the stand-in collective lives in the -DVP_DEBUG build only
                        ops._lib.call("vp_debug_occupy", 24, 3000000, torch.cuda.current_stream().cuda_stream)
            assert torch.equal(ops.gemm(a, w), ref)
            gu, act = ops.gemm_swiglu_fwd(x2, wgu)
            assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
            assert torch.equal(ops.gemm_tn(dyt, xt), tn_ref)
        torch.cuda.synchronize()
    finally:
        ops.set_dynamic(prev)
    assert torch.equal(ops.gemm(a, w), ref)


def test_w4_gemm_beside_small_kernels_on_a_second_stream(ops):
    """every instantiation of the one-wave-per-SIMD GEMM (lean / general / RMSNorm-fold epilogues) must give bit-identical results
    while a second stream keeps low-register kernels (element-wise, norm, fill: the kind that was placed beside a 464-register wave in round 4)
    and a collective-like CU hog flowing through the chip."""
    M, N, K = 8192, 4096, 1024
    a, w = dev(rnd(M, K, seed=190)), dev(rnd(N, K, scale=0.1, seed=191))
    bias, res = dev(rnd(N, seed=192)), dev(rnd(M, N, seed=193))
    Mt = M - 100                                                      # general variant's M tail
    solo = (ops.gemm(a, w), ops.gemm(a, w, bias=bias, residual=res, epi=ops.EPI_QUICK_GELU, force_generic=14),
            ops.gemm(a[:Mt], w, bias=bias, force_generic=14), *ops.gemm_sumsq(a, w, res))
    torch.cuda.synchronize()
    side, stop = torch.cuda.Stream(), torch.cuda.Event()
    sm = dev(rnd(96, 4096, seed=194))
    ones = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
    small = torch.empty(64 * 1024, device="cuda", dtype=torch.float32)
    for rnd_ in range(6):
        with torch.cuda.stream(side):
            for _ in range(60):                                       # ~600 small launches queued beside the GEMMs of this round
                ops.add(sm, sm)
                ops.rmsnorm_fwd(sm, ones, 1e-5)
                ops.act_fwd(sm, ops.EPI_GELU)
                ops.zero_(small)
            if rnd_ & 1:
                with ops._lib.debug_library():
                    ops._lib.call("vp_debug_occupy", 16, 2000000, torch.cuda.current_stream().cuda_stream)
        got = (ops.gemm(a, w), ops.gemm(a, w, bias=bias, residual=res, epi=ops.EPI_QUICK_GELU, force_generic=14),
               ops.gemm(a[:Mt], w, bias=bias, force_generic=14), *ops.gemm_sumsq(a, w, res))
        for i, (g_, s_) in enumerate(zip(got, solo)):
            assert torch.equal(g_, s_), f"round {rnd_}, output {i}: differs from the solo launch"
    torch.cuda.synchronize()


@pytest.mark.parametrize("kind", [1, 3])
def test_act(ops, kind):
    x, d = rnd(40, 64, seed=22), rnd(40, 64, seed=23)
    xr = x.float().requires_grad_(True)
    ref = F.gelu(xr) if kind == 1 else F.relu(xr)
    ref.backward(d.float())
    close(ops.act_fwd(dev(x), kind), ref, what="act fwd")
    close(ops.act_bwd(dev(d), dev(x), kind), xr.grad, what="act bwd")


def test_add_colsum_cast(ops):
    a, b = rnd(70, 64, seed=24), rnd(70, 64, seed=25)
    close(ops.add(dev(a), dev(b)), a.float() + b.float(), what="add")
    close(ops.colsum(dev(a)), a.float().sum(0), rtol=1e-3, what="colsum")
    x32 = torch.randn(1000)
    assert torch.equal(ops.cast_to_bf16(x32.cuda()).cpu(), x32.to(BF))
    close(ops.sum_f32(x32.cuda(), 0.5), x32.sum()[None] * 0.5, rtol=1e-4, atol=1e-3, what="sum")
    dst = dev(rnd(20, 96, seed=26)); src = dev(rnd(20, 64, seed=27))
    ref = dst.clone().float().cpu(); ref[:, 16:80] += src.float().cpu()
    ops.add2d_(dst[:, 16:80], src)
    close(dst, ref, what="add2d")


def test_scatter_rows_bf16_to_f32(ops):
    """vp_scatter_rows_bf16_to_f32: bf16 rows widened into chosen rows of an fp32 matrix (the reference's `logits.float()`, ola_llama.py:122);
    skipped rows (idx < 0) and rows nobody writes stay untouched; idx = None is the row-for-row cast.  Exact (a widening)."""
    src = rnd(37, 1000, seed=301)
    idx = torch.randperm(64, generator=torch.Generator().manual_seed(5))[:37].to(torch.int32)
    idx[5] = -1
    dst = torch.full((64, 1000), 7.0, device="cuda")
    ops.scatter_rows_to_f32(dev(src), dev(idx), dst)
    ref = torch.full((64, 1000), 7.0)
    for r in range(37):
        if idx[r] >= 0:
            ref[idx[r]] = src[r].float()
    assert torch.equal(dst.cpu(), ref)
    d2 = torch.empty(40, 1000, device="cuda")
    ops.scatter_rows_to_f32(dev(src), None, d2[2:39])                      # row-for-row into a slice
    assert torch.equal(d2[2:39].cpu(), src.float())
    with pytest.raises(RuntimeError):
        ops.scatter_rows_to_f32(dev(rnd(4, 12, seed=302)), None, torch.empty(4, 12, device="cuda"))      # H % 8 != 0


def test_gather_and_gather_sum(ops):
    H = 64
    s0, s1, s2 = rnd(10, H, seed=28), rnd(7, H, seed=29), rnd(5, H, seed=30)
    kind = torch.tensor([0, 1, -1, 2, 0, 1, 2, -1, 0], dtype=torch.int32)
    row = torch.tensor([3, 6, 0, 4, 9, 0, 1, 0, 0], dtype=torch.int32)
    out = torch.empty(len(kind), H, device="cuda", dtype=BF)
    ops.gather_rows([dev(s0), dev(s1), dev(s2)], kind.cuda(), row.cuda(), H, out)
    srcs = [s0, s1, s2]
    ref = torch.stack([srcs[k][r] if k >= 0 else torch.zeros(H, dtype=BF) for k, r in zip(kind.tolist(), row.tolist())])
    assert torch.equal(out.cpu(), ref)
    idx = torch.tensor([[0, 1, 2], [3, -1, 9], [4, 4, 4]], dtype=torch.int32)
    o2 = torch.empty(3, H, device="cuda", dtype=torch.float32)
    ops.gather_sum_rows(dev(s0), idx.cuda().flatten(), 3, 0.5, o2)
    ref2 = torch.stack([sum(s0[i].float() for i in r if i >= 0) * 0.5 for r in idx.tolist()])
    close(o2, ref2, rtol=1e-4, what="gather_sum")


def test_adamw(ops):
    n = 5000
    p0, g = torch.randn(n), torch.randn(n)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    sh = torch.empty(n, device="cuda", dtype=BF)
    for step in (1, 2, 3):
        pr.grad = g.clone() * step
        opt.step()
        ops.adamw_(p, (g * step).cuda(), m, v, sh, 1e-2, 0.9, 0.999, 1e-8, 0.1, step)
    close(p, pr.detach(), rtol=1e-5, atol=1e-6, what="adamw")
    assert torch.equal(sh.cpu(), p.cpu().to(BF))


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, causal, kv_len=None, window=0):
    """q [B,Sq,Hq,D], k,v [B,Skv,Hkv,D] fp32 -> o [B,Sq,Hq,D] with HF-eager semantics."""
    B, Sq, Hq, D = q.shape
    Skv, Hkv = k.shape[1], k.shape[2]
    rep = Hq // Hkv
    qq = q.transpose(1, 2)
    kk = k.transpose(1, 2).repeat_interleave(rep, dim=1)
    vv = v.transpose(1, 2).repeat_interleave(rep, dim=1)
    s = qq @ kk.transpose(-1, -2) / math.sqrt(D)
    i = torch.arange(Sq)[:, None] + (Skv - Sq)
    j = torch.arange(Skv)[None, :]
    allow = torch.ones(Sq, Skv, dtype=torch.bool)
    if causal:
        allow &= j <= i
    if window > 0:
        allow &= j > i - window
    allow = allow[None, None].expand(B, 1, Sq, Skv).clone()
    if kv_len is not None:
        for b in range(B):
            allow[b, :, :, kv_len[b]:] = False
    s = s.masked_fill(~allow, float("-inf"))
    return (torch.softmax(s, -1) @ vv).transpose(1, 2)


@pytest.mark.parametrize("B,Hq,Hkv,Sq,Skv,D,causal", [
    (2, 4, 2, 128, 128, 128, True),       # Llama-style GQA causal
    (1, 4, 4, 200, 200, 64, True),        # ragged length causal
    (2, 4, 4, 577, 577, 64, False),       # ViT
    (2, 4, 4, 1, 100, 32, False),         # gen head: one query
    (2, 4, 4, 70, 210, 32, False),        # resampler cross attention
    (1, 2, 2, 150, 150, 96, True),        # Phi-3 head dim
    (1, 8, 2, 320, 320, 128, True),
])
def test_attention_fwd_bwd(ops, B, Hq, Hkv, Sq, Skv, D, causal):
    q, k, v, do = rnd(B, Sq, Hq, D, seed=31), rnd(B, Skv, Hkv, D, seed=32), rnd(B, Skv, Hkv, D, seed=33), rnd(B, Sq, Hq, D, seed=34)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qr, kr, vr, causal)
    ref.backward(do.float())
    qd, kd, vd = dev(q), dev(k), dev(v)
    o, lse = ops.attn_fwd(qd, kd, vd, causal)
    close(o, ref, what="attn fwd")
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, o, lse, dev(do), causal)
    close(dq, qr.grad, rtol=3e-2, what="attn dq")
    close(dk, kr.grad, rtol=3e-2, what="attn dk")
    close(dv, vr.grad, rtol=3e-2, what="attn dv")


def test_attention_fused_qkv_views_and_kvlen(ops):
    B, S, Hq, Hkv, D = 2, 96, 4, 2, 64
    qkv = rnd(B, S, (Hq + 2 * Hkv) * D, seed=35)
    kv_len = [96, 61]
    q = qkv[..., :Hq * D].reshape(B, S, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].reshape(B, S, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].reshape(B, S, Hkv, D)
    ref = attn_ref(q.float(), k.float(), v.float(), True, kv_len=kv_len)
    d = dev(qkv)
    qd = d[..., :Hq * D].view(B, S, Hq, D)
    kd = d[..., Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
    vd = d[..., (Hq + Hkv) * D:].view(B, S, Hkv, D)
    o, lse = ops.attn_fwd(qd, kd, vd, True, kv_len=torch.tensor(kv_len, dtype=torch.int32).cuda())
    close(o, ref, what="attn fused-qkv kvlen")


def test_attention_sliding_window(ops):
    B, S, H, D = 1, 200, 2, 32
    q, k, v = rnd(B, S, H, D, seed=36), rnd(B, S, H, D, seed=37), rnd(B, S, H, D, seed=38)
    ref = attn_ref(q.float(), k.float(), v.float(), True, window=70)
    o, _ = ops.attn_fwd(dev(q), dev(k), dev(v), True, window=70)
    close(o, ref, what="attn window")


# ------------------------------------------------------------------------------------------------ losses
def test_ce(ops):
    rows, V = 37, 1000
    lg = rnd(rows, V, scale=2.0, seed=39)
    labels = torch.randint(0, V, (rows,))
    labels[::5] = -100
    n = int((labels != -100).sum())
    lr = lg.float().requires_grad_(True)
    ref = F.cross_entropy(lr, labels, ignore_index=-100, reduction="sum")
    (ref / n).backward()
    lgd = dev(lg)
    rl = ops.ce_fwd_bwd(lgd, labels.cuda(), 1.0 / n)
    close(rl.sum()[None], ref[None], rtol=1e-4, atol=1e-3, what="ce loss")
    close(lgd, lr.grad, rtol=2e-2, atol=2e-3 * float(lr.grad.abs().max()), what="ce dlogits")


@pytest.mark.parametrize("B,world,rank,shape,mask", [(3, 1, 0, (3, 40, 1024), [1., 0., 1.]), (2, 4, 2, (2, 96, 24), [1., 1.]),
                                                     (8, 1, 0, (8, 1, 1024), [1.] * 8), (4, 2, 1, (4, 1536, 6, 6), [1., 1., 0., 1.]),
                                                     # the reference's pretrain.sh runs 32 per device on 8 devices (Bw = 256); odd local
                                                     # batches; feature length not a multiple of the 32-wide k-step; several target chunks (Bw > 128)
                                                     (32, 8, 5, (32, 40, 64), [1.] * 31 + [0.]), (12, 3, 1, (12, 1000), [1.] * 12),
                                                     (20, 4, 3, (20, 7, 24), [0.5] * 20), (8, 8, 7, (8, 576, 64), [1.] * 8),
                                                     (64, 16, 9, (64, 264), [1.] * 64)])
def test_emb_loss(ops, B, world, rank, shape, mask):
    from oracle import visper_oracle as O
    D = math.prod(shape[1:])
    pred = rnd(*shape, scale=1.3, seed=40)
    tg_all = rnd(B * world, D, seed=41)
    tgt = tg_all[rank * B:(rank + 1) * B].reshape(shape)
    pr = pred.float().requires_grad_(True)
    ls = torch.tensor(2.0, requires_grad=True)
    gathered = F.normalize(tg_all.float(), dim=-1)
    e, s1, c = O.emb_loss(pr, torch.tensor(mask), tgt.float(), ls, 0.3, rank=rank, gathered_targets=gathered)
    (e * 0.5).backward()
    out3, coef = ops.emb_loss_fwd(dev(pred).reshape(B, D), dev(tg_all), torch.tensor(mask).cuda(), torch.tensor([2.0]).cuda(), 0.3,
                                  rank=rank)
    close(out3, torch.stack([e, s1, c]).detach(), rtol=2e-3, atol=1e-5, what="emb loss fwd")
    dp = ops.emb_loss_bwd(dev(pred).reshape(B, D), dev(tg_all), coef, 0.5, rank=rank)
    close(dp.reshape(shape), pr.grad, rtol=3e-2, atol=2e-2 * float(pr.grad.abs().max()), what="emb loss dpred")
    close(coef[-1:] * 0.5, ls.grad[None], rtol=5e-3, atol=1e-6, what="dlogit_scale")
    # one launch, deterministic tree reduction: bitwise identical on a second call (and the ticket counters re-armed themselves)
    out3b, coefb = ops.emb_loss_fwd(dev(pred).reshape(B, D), dev(tg_all), torch.tensor(mask).cuda(), torch.tensor([2.0]).cuda(), 0.3,
                                    rank=rank)
    assert torch.equal(out3, out3b) and torch.equal(coef, coefb)


def test_emb_loss_full_size_world8(ops):
    """K11 at the config-2 seg shape (D = 1536*576) with 8 ranks' gathered targets: statistics against fp64 on a feature subsample-free
    full reduction (torch on the GPU is only the checker here), no-contrastive variant, and the shapes the C ABI refuses."""
    B, Bw, D, rank = 8, 64, 1536 * 576, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    pred = (torch.randn(B, D, device="cuda", generator=g) * 1.3).to(torch.bfloat16)
    tgt = torch.randn(Bw, D, device="cuda", generator=g).to(torch.bfloat16)
    mask = torch.ones(B, device="cuda")
    out3, coef = ops.emb_loss_fwd(pred, tgt, mask, torch.tensor([2.0]).cuda(), 0.3, rank=rank)
    p64, t64 = pred.double(), tgt.double()
    own = t64[rank * B:(rank + 1) * B]
    d = (p64 - own).abs()
    sl1 = torch.where(d < 1, 0.5 * d * d, d - 0.5).mean()
    z = (F.normalize(p64, dim=-1) @ F.normalize(t64, dim=-1).t()) * min(math.exp(2.0), 100.0)
    con = 0.3 * F.cross_entropy(z, torch.arange(B, device="cuda") + rank * B)
    close(out3.double().cpu(), torch.stack([sl1 + con, sl1, con]).cpu(), rtol=1e-4, atol=1e-6, what="emb loss full size")
    out3n, coefn = ops.emb_loss_fwd(pred, tgt, mask, None, 0.3, rank=rank)
    assert float(out3n[2]) == 0.0 and abs(float(out3n[1]) - float(sl1)) < 1e-4 * float(sl1) and float(coefn[B:-1].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.emb_loss_fwd(pred[:, :1004].contiguous(), tgt[:, :1004].contiguous(), mask, None, 0.3)       # D % 8 != 0
    with pytest.raises(RuntimeError):
        ops.emb_loss_fwd(pred, tgt[:4], mask, None, 0.3)                                               # Bw < B


def test_emb_loss_concurrent_streams(ops):
    """The loss forward's ticket counters live in a CALLER-owned block (include/visper_hip.h `counters`; round 6: the library owns no device
    memory); ops._loss_counters keeps one zeroed block per stream, so launches that overlap on several streams never share tickets.
    12 streams run the single-launch forward
    concurrently, 6 rounds each on its own data; every result must equal the bits of the same call made alone on the default stream, and
    a later default-stream call must still be exact (no counter left armed)."""
    B, Bw, D = 8, 16, 576 * 256
    g = torch.Generator(device="cuda").manual_seed(9)
    preds = [(torch.randn(B, D, device="cuda", generator=g) * 1.3).to(torch.bfloat16) for _ in range(12)]
    tgts = [torch.randn(Bw, D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(12)]
    mask, ls = torch.ones(B, device="cuda"), torch.tensor([2.0], device="cuda")
    alone = [ops.emb_loss_fwd(p, t, mask, ls, 0.3, rank=1) for p, t in zip(preds, tgts)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(12)]
    got = [[] for _ in range(12)]
    for _ in range(6):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                got[i].append(ops.emb_loss_fwd(preds[i], tgts[i], mask, ls, 0.3, rank=1))
    torch.cuda.synchronize()
    for i in range(12):
        for out3, coef in got[i]:
            assert torch.equal(out3, alone[i][0]) and torch.equal(coef, alone[i][1]), i
    again = ops.emb_loss_fwd(preds[0], tgts[0], mask, ls, 0.3, rank=1)
    assert torch.equal(again[0], alone[0][0]) and torch.equal(again[1], alone[0][1])


def test_emb_loss_more_streams_than_counter_sets(ops):
    """ADVICE r3 (the library-side registry of 32 counter sets is gone in round 6: the block is the caller's): 40 distinct streams, some created
    and dropped in between, call the loss twice each, every one with its own caller-owned counter block.  Every result must equal the bits of the
    same call on the default stream, also when earlier streams come back."""
    B, Bw, D = 8, 8, 576 * 64
    g = torch.Generator(device="cuda").manual_seed(10)
    pred = (torch.randn(B, D, device="cuda", generator=g) * 1.1).to(torch.bfloat16)
    tgt = torch.randn(Bw, D, device="cuda", generator=g).to(torch.bfloat16)
    mask, ls = torch.ones(B, device="cuda"), torch.tensor([2.0], device="cuda")
    ref = ops.emb_loss_fwd(pred, tgt, mask, ls, 0.3)
    torch.cuda.synchronize()
    keep = []
    for i in range(40):
        st = torch.cuda.Stream()
        if i % 3:
            keep.append(st)
        with torch.cuda.stream(st):
            for _ in range(2):
                out3, coef = ops.emb_loss_fwd(pred, tgt, mask, ls, 0.3)
        st.synchronize()
        assert torch.equal(out3, ref[0]) and torch.equal(coef, ref[1]), i
    for st in keep[:8]:                                                       # the earliest streams again: their sets were re-assigned meanwhile
        with torch.cuda.stream(st):
            out3, coef = ops.emb_loss_fwd(pred, tgt, mask, ls, 0.3)
        st.synchronize()
        assert torch.equal(out3, ref[0]) and torch.equal(coef, ref[1])
    again = ops.emb_loss_fwd(pred, tgt, mask, ls, 0.3)
    assert torch.equal(again[0], ref[0]) and torch.equal(again[1], ref[1])


def test_dpt_conv_helpers(ops):
    """conv.hip (frozen DPT decoder, da_v2_head.py:182-321): im2col3x3 + GEMM == F.conv2d, GEMM + pixel shuffle ==
    F.conv_transpose2d(k = stride), bilinear align_corners=True, per-image min-max normalisation."""
    B, H, W, C, Co = 2, 12, 10, 16, 24
    x = rnd(B, H, W, C, seed=40)
    xn = x.float().permute(0, 3, 1, 2)
    for stride, relu_in in ((1, False), (2, False), (1, True)):
        w = rnd(Co, C, 3, 3, seed=41) * 0.2
        b = rnd(Co, seed=42)
        ref = F.conv2d(F.relu(xn) if relu_in else xn, w.float(), b.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
        wm = dev(w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous())
        got = ops.conv3x3_nhwc(dev(x), wm, bias=dev(b), stride=stride, relu_in=relu_in)
        close(got, ref, what=f"conv3x3 s{stride} relu{relu_in}")
    for k in (2, 4):
        wt = rnd(C, Co, k, k, seed=43) * 0.2
        bt = rnd(Co, seed=44)
        ref = F.conv_transpose2d(xn, wt.float(), bt.float(), stride=k).permute(0, 2, 3, 1)
        wm = dev(wt.permute(2, 3, 1, 0).reshape(k * k * Co, C).contiguous())
        close(ops.conv_transpose_nhwc(dev(x), wm, dev(bt.repeat(k * k)), k), ref, what=f"conv_transpose k{k}")
    for (Ho, Wo) in ((24, 20), (17, 31), (12, 10)):
        ref = F.interpolate(xn, size=(Ho, Wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        close(ops.bilinear_nhwc(dev(x), Ho, Wo), ref, rtol=1e-2, what=f"bilinear {Ho}x{Wo}")
    d = rnd(3, 1000, seed=45).abs()
    mn, mx = d.float().amin(1, keepdim=True), d.float().amax(1, keepdim=True)
    close(ops.minmax_norm(dev(d)), (d.float() - mn) / (mx - mn), what="minmax")


@pytest.mark.parametrize("B,H,W,C", [(2, 12, 16, 16), (1, 5, 13, 24), (2, 3, 7, 8), (1, 24, 24, 64)])
def test_dwconv7x7_nhwc(ops, B, H, W, C):
    """vp_dwconv7x7_nhwc (timm ConvNeXtBlock.conv_dw, clip_convnext_encoder.py:161-165) == F.conv2d(groups = C, padding 3) in fp32, rounded once:
    widths that are / are not multiples of the kernel's 8-pixel strip, maps smaller than the 7 x 7 window, and an exact fp32 replay of the
    kernel's accumulation order (bias, then taps in (dy, dx) order, fused multiply-adds) on a sample of outputs."""
    x = rnd(B, H, W, C, seed=50)
    w = rnd(C, 1, 7, 7, seed=51) * 0.2
    b = rnd(C, seed=52)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), padding=3, groups=C).permute(0, 2, 3, 1)
    wt = dev(w.reshape(C, 49).t().contiguous())                       # tap-major [49, C]
    got = ops.dwconv7x7_nhwc(dev(x), wt, dev(b))
    close(got, ref, rtol=1e-2, what=f"dwconv7x7 {B}x{H}x{W}x{C}")
    again = ops.dwconv7x7_nhwc(dev(x), wt, dev(b))
    assert torch.equal(got, again)
    # the error against fp32 math is the one bf16 rounding of the output
    err = (got.float().cpu() - ref).abs()
    assert float((err / (ref.abs() + 1e-3)).max()) < 8e-3


def test_param_store_adamw_groups_schedule_clipping():
    """ParamStore.adamw_step (grouped fused launches + schedule multiplier + global-norm clipping) == torch.optim.AdamW with the
    reference trainer's parameter groups, transformers' cosine schedule and clip_grad_norm_."""
    from collections import OrderedDict
    from transformers import get_cosine_schedule_with_warmup
    from visper_lm_amd import optim
    from visper_lm_amd.engine import ParamStore
    shapes = OrderedDict([("image_gen_heads.0.projector.proj_in.weight", (24, 40)), ("image_gen_heads.0.projector.proj_in.bias", (24,)),
                          ("image_gen_heads.0.projector.norm_out.weight", (24,)), ("gen_logit_scale", ()),
                          ("model.mm_projector.0.weight", (40, 16)), ("model.mm_projector.0.bias", (40,)), ("model.special_gen_tokens", (8, 40))])
    ps = ParamStore(shapes, "cuda")
    ref = OrderedDict((k, torch.nn.Parameter(rnd(*s, seed=60 + i).float() if len(s) else torch.tensor(2.0))) for i, (k, s) in enumerate(shapes.items()))
    for k, v in ref.items():
        ps.p(k).copy_(v.detach().reshape(ps.p(k).shape))
    base_lr, wd, plr, total, mx = 1e-2, 0.1, 3e-3, 12, 0.7
    gr = optim.param_groups(shapes.keys(), wd, plr)
    opt = torch.optim.AdamW([dict(params=[ref[k]], weight_decay=gr[k][1], lr=(base_lr if gr[k][0] is None else gr[k][0])) for k in shapes],
                            lr=base_lr, betas=(0.9, 0.999), eps=1e-8)
    nw = optim.warmup_steps(total, 0.25)
    sch = get_cosine_schedule_with_warmup(opt, nw, total)
    for step in range(6):
        for i, (k, v) in enumerate(ref.items()):
            g = (rnd(*shapes[k], seed=100 + 10 * step + i).float() if len(shapes[k]) else torch.tensor(0.3 * (step + 1))) * (2.0 if step % 2 else 0.2)
            v.grad = g.clone()
            ps.g(k).copy_(g.reshape(ps.g(k).shape))
        torch.nn.utils.clip_grad_norm_(list(ref.values()), mx)
        opt.step(); 
        ps.adamw_step(base_lr, weight_decay=wd, mm_projector_lr=plr, lr_mult=optim.cosine_with_warmup(step, total, nw), max_grad_norm=mx)
        sch.step()
    for k, v in ref.items():
        close(ps.p(k).view(v.shape), v.detach(), rtol=2e-5, atol=2e-6, what=f"adamw {k}")
        close(ps.w(k).view(v.shape), v.detach(), rtol=1e-2, what=f"bf16 shadow {k}")


def test_ift_support_kernels(ops):
    """Kernels the IFT stage adds: [gate | up]-halves SwiGLU layout, RMSNorm weight gradient, embedding scatter-add, sum of squares."""
    M, Fd = 37, 72
    gu, d = rnd(M, 2 * Fd, seed=70), rnd(M, Fd, seed=71)
    gr = gu.float().requires_grad_(True)
    ref = F.silu(gr[:, :Fd]) * gr[:, Fd:]
    ref.backward(d.float())
    close(ops.swiglu_fwd(dev(gu), interleaved=False), ref, what="swiglu fwd halves")
    close(ops.swiglu_bwd(dev(d), dev(gu), interleaved=False), gr.grad, what="swiglu bwd halves")
    # RMSNorm weight gradient: y = x * rstd * w  ->  dw = sum_rows dy * x * rstd
    Mr, H = 300, 256
    x, dy, w = rnd(Mr, H, seed=72), rnd(Mr, H, seed=73), (rnd(H, seed=74) * 0.1 + 1.0)
    xf = x.float()
    wf = w.float().requires_grad_(True)
    rstd = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    (xf * rstd * wf).backward(dy.float())
    close(ops.rmsnorm_bwd_w(dev(dy), dev(x), dev(rstd.reshape(-1).contiguous())), wf.grad, rtol=1e-3, what="rmsnorm dw")
    # embedding gradient: dst[idx[r]] += src[r], idx < 0 skipped
    V, Hh, n = 50, 64, 400
    src = rnd(n, Hh, seed=75)
    idx = torch.randint(-1, V, (n,), generator=torch.Generator().manual_seed(76)).to(torch.int32)
    want = torch.zeros(V, Hh)
    keep = idx >= 0
    want.index_add_(0, idx[keep].long(), src[keep].float())
    got = ops.scatter_add_rows_(torch.zeros(V, Hh, device="cuda"), dev(src), idx.cuda())
    close(got, want, rtol=1e-4, atol=1e-4, what="scatter_add_rows")
    v = torch.randn(100003)
    assert abs(float(ops.sumsq(v.cuda())) - float((v.double() ** 2).sum())) < 1e-3 * float((v.double() ** 2).sum())


def test_gemm_random_shape_sweep(ops):
    """Seeded sweep over the dispatcher's corner cases: ragged M/N (partial 256 / 128 tiles), K not a multiple of 64 (generic
    kernel), strided A / C views, odd bias offsets (unaligned bias pointer -> general epilogue), every epilogue, fp32 output."""
    rng = np.random.RandomState(1234)
    for case in range(40):
        M = int(rng.choice([1, 7, 64, 130, 255, 256, 300, 513, 1024, 2200]))
        N = int(rng.choice([8, 24, 64, 136, 256, 264, 520, 1032]))
        K = int(rng.choice([8, 40, 64, 128, 192, 200, 512]))
        epi = int(rng.choice([0, 0, 1, 2, 3]))
        use_bias, use_res, f32, strided = rng.rand() < 0.6, rng.rand() < 0.5, rng.rand() < 0.2, rng.rand() < 0.3
        force = int(rng.choice([0, 0, 0, 2, 3, 7]))
        if K % 64 != 0 and force in (2, 3, 7):
            force = 0
        a_full = rnd(M, K + (16 if strided else 0), seed=1000 + case)
        a = a_full[:, 8:8 + K] if strided else a_full
        w = rnd(N, K, scale=0.1, seed=2000 + case)
        bias_buf = rnd(N + 3, seed=3000 + case)
        b = bias_buf[3:3 + N] if (use_bias and case % 3 == 0) else (bias_buf[:N] if use_bias else None)   # odd offset: 2-byte aligned only
        r = rnd(M, N, seed=4000 + case) if (use_res and not f32) else None
        x = a.float() @ w.float().t() + (b.float() if b is not None else 0.0)
        if not f32:
            x = x.to(BF).float()
            if epi == 1:
                x = F.gelu(x).to(BF).float()
            elif epi == 2:
                x = (x * torch.sigmoid(1.702 * x)).to(BF).float()
            elif epi == 3:
                x = F.relu(x)
            if r is not None:
                x = (x + r.float()).to(BF).float()
        elif epi != 0:
            continue                                            # activations are a bf16-output feature
        a_dev = dev(a_full)[:, 8:8 + K] if strided else dev(a)
        b_dev = None if b is None else (dev(bias_buf)[3:3 + N] if (use_bias and case % 3 == 0) else dev(bias_buf)[:N])
        got = ops.gemm(a_dev, dev(w), bias=b_dev, residual=None if r is None else dev(r), epi=epi, out_f32=f32, force_generic=force)
        close(got, x, what=f"gemm sweep #{case} M{M} N{N} K{K} epi{epi} bias{use_bias} res{r is not None} f32{f32} strided{strided} force{force}")


@pytest.mark.parametrize("B,Hq,Hkv,Sq,Skv,causal,window,kvl", [
    (2, 4, 2, 77, 77, True, 0, None),            # ragged (not a multiple of the 32-row DMA tiles), GQA
    (1, 4, 4, 200, 333, False, 0, None),         # cross attention, Skv != Sq
    (1, 4, 1, 130, 190, True, 0, None),          # causal with offset Skv - Sq, 4 q heads per kv head
    (2, 2, 2, 161, 161, True, 50, None),         # sliding window
    (2, 4, 2, 96, 96, True, 0, [96, 41]),        # right-padded batch: keys >= kv_len masked, their dK / dV exactly zero
    (1, 2, 2, 31, 31, True, 0, None),            # shorter than one tile
    (1, 2, 1, 257, 257, False, 0, [200]),        # non-causal + padding, one key past two 128-key blocks
])
@pytest.mark.parametrize("D", [128, 96])
def test_attention_d128_dma_kernels_edges(ops, B, Hq, Hkv, Sq, Skv, causal, window, kvl, D):
    """The DMA-ring kernels (dK/dV, dQ; D = 128 and, with 3 of 4 k-steps / 6 of 8 feature blocks, Phi-3's D = 96) and the matching forward on
    shapes that exercise clamped tile rows, the causal offset, windows and kv_len masking, forward and backward against fp32 attention."""
    q, k, v, do = rnd(B, Sq, Hq, D, seed=81), rnd(B, Skv, Hkv, D, seed=82), rnd(B, Skv, Hkv, D, seed=83), rnd(B, Sq, Hq, D, seed=84)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qr, kr, vr, causal, kv_len=kvl, window=window)
    ref.backward(do.float())
    kv = None if kvl is None else torch.tensor(kvl, dtype=torch.int32).cuda()
    qd, kd, vd = dev(q), dev(k), dev(v)
    o, lse = ops.attn_fwd(qd, kd, vd, causal, window=window, kv_len=kv)
    close(o, ref, what="attn fwd")
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, o, lse, dev(do), causal, window=window, kv_len=kv)
    close(dq, qr.grad, rtol=3e-2, what="attn dq")
    close(dk, kr.grad, rtol=3e-2, what="attn dk")
    close(dv, vr.grad, rtol=3e-2, what="attn dv")
    if kvl is not None:
        for b, n in enumerate(kvl):
            assert float(dk[b, n:].float().abs().max() if n < Skv else 0.0) == 0.0
            assert float(dv[b, n:].float().abs().max() if n < Skv else 0.0) == 0.0


@pytest.mark.parametrize("S,window,kvl", [(700, 0, None), (700, 300, [700, 512]), (257, 64, None)])
def test_attention_d96_fused_qkv_views_fwd_bwd(ops, S, window, kvl):
    """Phi-3's layout (ola_phi3.py -> HF Phi3Attention: ONE qkv_proj output, 32 MHA heads x 96): q / k / v are strided views of one [B, S, 3 H D]
    tensor whose LAST row / head ends the allocation (the D = 96 kernels keep 128-wide LDS rows: their DMA lanes for the 4 unused chunks must
    stay inside the tensor), dq / dk / dv are written into views of one buffer.  Forward + backward against fp32 attention, causal, with a sliding
    window longer and shorter than a tile and a right-padded batch."""
    B, H, D = 2, 4, 96
    qkv = rnd(B, S, 3 * H * D, seed=131)
    do = rnd(B, S, H, D, seed=132)
    q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].reshape(B, S, H, D) for i in range(3))
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qr, kr, vr, True, kv_len=kvl, window=window)
    ref.backward(do.float())
    d = dev(qkv)
    qd, kd, vd = (d[..., i * H * D:(i + 1) * H * D].view(B, S, H, D) for i in range(3))
    kv = None if kvl is None else torch.tensor(kvl, dtype=torch.int32).cuda()
    o, lse = ops.attn_fwd(qd, kd, vd, True, window=window, kv_len=kv)
    close(o, ref, what="attn d96 fwd")
    dqkv = torch.full_like(d, float("nan"))
    dq, dk, dv = (dqkv[..., i * H * D:(i + 1) * H * D].view(B, S, H, D) for i in range(3))
    ops.attn_bwd(qd, kd, vd, o, lse, dev(do), True, window=window, kv_len=kv, dq=dq, dk=dk, dv=dv)
    assert torch.isfinite(dqkv.float()).all()                                # every element of the three views was written
    close(dq, qr.grad, rtol=3e-2, what="attn d96 dq")
    close(dk, kr.grad, rtol=3e-2, what="attn d96 dk")
    close(dv, vr.grad, rtol=3e-2, what="attn d96 dv")


@pytest.mark.parametrize("causal", [True, False])
def test_attention_d128_rescale_branch_mid_sequence(ops, causal):
    """The deferred running max of the D = 128 forward kernels (O / l are only rescaled when a tile's max exceeds the running max by > 2^8):
    on random inputs that branch fires in the first tile only, so a stale or half-applied rescale later in the sequence would go unnoticed.
    Here a few keys are aligned with a few query rows so that their scores jump by ~30 and then by another ~16 log2 units in LATER tiles
    (key positions in both key blocks of a 64-key tile and in both lane halves of the 32x32 layout; query rows in different waves), the rest
    of the rows stay ordinary.  Forward vs fp32 softmax over the whole tensor, backward (which consumes the forward's lse) vs autograd."""
    B, Hq, Hkv, S, D = 1, 4, 2, 512, 128
    q, k, v, do = rnd(B, S, Hq, D, seed=301), rnd(B, S, Hkv, D, seed=302), rnd(B, S, Hkv, D, seed=303), rnd(B, S, Hq, D, seed=304)
    u = torch.ones(D) / math.sqrt(D)
    for j, mag in ((130, 16.0), (140, 16.0), (170, 16.0), (300, 24.0), (450, 32.0)):
        k[0, j, :, :] = (mag * u).to(BF)
        v[0, j, :, :] = (v[0, j, :, :].float() * 3.0).to(BF)
    for r in (5, 135, 200, 260, 310, 400, 460, 511):
        q[0, r, :, :] = (16.0 * u + 0.5 * q[0, r, :, :].float()).to(BF)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qr, kr, vr, causal)
    ref.backward(do.float())
    qd, kd, vd = dev(q), dev(k), dev(v)
    o, lse = ops.attn_fwd(qd, kd, vd, causal)
    close(o, ref, what="attn fwd (rescale branch)")
    # the aligned rows really are dominated by a late key (otherwise this test exercises nothing)
    with torch.no_grad():
        sc = (qr[0, 511, 0] @ kr[0, :, 0].T) / math.sqrt(D)
        assert float(sc.max() - sc[:128].max()) * 1.4427 > 16.0
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, o, lse, dev(do), causal)
    close(dq, qr.grad, rtol=3e-2, what="attn dq (rescale branch)")
    close(dk, kr.grad, rtol=3e-2, what="attn dk (rescale branch)")
    close(dv, vr.grad, rtol=3e-2, what="attn dv (rescale branch)")


def test_attention_forward_variants_by_env():
    """The generic 16-row forward kernel at D = 128 (VP_ATTN_FWDM=0; the launcher reads its switch once per process, so it runs in a child
    process) re-runs the D = 128 edge cases and the decoder-shaped case of this file: the fall-back the 32x32x16 kernel is A/B-ed against."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in ({"VP_ATTN_FWDM": "0"},):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                            "test_attention_d128_dma_kernels_edges or test_attention_fwd_bwd or test_attention_fused_qkv_views_and_kvlen or test_attention_sliding_window "
                            "or test_attention_d128_rescale_branch_mid_sequence"],
                           capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, **env))
        assert r.returncode == 0 and " passed" in r.stdout, (env, r.stdout[-1500:], r.stderr[-500:])


def test_attention_backward_variants_by_env():
    """VP_ATTN_BWD64 (read once per process -> child processes): 1 = round 5's one-wave-per-SIMD dQ + dK/dV kernels (attention_bwd64.h), 2 = round 4's
    dQ kernel (writing the statistics planes) in front of round 5's dK/dV kernel, 0 = round 4's pair.  Every backward test of this file (edges, GQA,
    kv_len, windows, D = 96, fused RoPE^T) must pass under each, whichever is the default."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("0", "1", "2"):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                            "test_attention_d128_dma_kernels_edges or test_attention_fwd_bwd or test_attention_fused_qkv_views_and_kvlen or test_attention_sliding_window "
                            "or test_attention_d96_fused_qkv_views_fwd_bwd or test_attn_bwd_fused_rope or test_attention_d128_rescale_branch_mid_sequence"],
                           capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, VP_ATTN_BWD64=mode))
        assert r.returncode == 0 and " passed" in r.stdout, (mode, r.stdout[-1500:], r.stderr[-500:])


def test_attention_with_additive_biases(ops):
    """vp_attn_fwd_bias (Swin window attention): per-head relative-position bias + per-window shift mask (-100 entries, HF
    modeling_swin.py get_attn_mask), 144-token windows, D = 32, batch = images x windows."""
    nW, imgs, H, N, D = 4, 2, 3, 144, 32
    B = nW * imgs
    q, k, v = rnd(B, N, H, D, seed=91), rnd(B, N, H, D, seed=92), rnd(B, N, H, D, seed=93)
    bh = rnd(H, N, N, seed=94).float() * 2.0
    mask = torch.zeros(nW, N, N)
    mask[1, :, 72:] = -100.0
    mask[1, 72:, :72] = -100.0
    mask[1, 72:, 72:] = 0.0
    mask[3, :60, 60:] = -100.0
    s = (q.float().transpose(1, 2) @ k.float().transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)          # [B,H,N,N]
    s = s + bh[None] + mask.repeat(imgs, 1, 1)[:, None]
    ref = (torch.softmax(s, -1) @ v.float().transpose(1, 2)).transpose(1, 2)
    got = ops.attn_fwd_bias(dev(q), dev(k), dev(v), bias_h=dev(bh).contiguous(), bias_b=dev(mask).contiguous())
    close(got, ref, what="attn bias_h + bias_b")
    close(ops.attn_fwd_bias(dev(q), dev(k), dev(v), bias_h=dev(bh).contiguous()), (torch.softmax(s - mask.repeat(imgs, 1, 1)[:, None], -1) @ v.float().transpose(1, 2)).transpose(1, 2), what="attn bias_h only")


@pytest.mark.gpu
def test_clock_probe(ops):
    """The measurement aid behind roofline.clock (include/visper_hip_debug.h): in-kernel stamps give plausible per-XCD numbers."""
    import bench
    ck = bench.clock_probe(torch.device("cuda"), n=6)
    assert 800 < ck["shader_clock_mhz"] <= 2500 and len(ck["per_xcd_mhz"]) == 8, ck
    assert 0.5 < ck["mfma_issue_util_in_k_loop"] <= 1.0, ck


def test_emb_loss_multi_task_launch_equals_single_launches(ops):
    """vp_emb_loss_{fwd,bwd}_multi: the three distillation heads of a step (different D, same B / Bw / rank) in ONE launch each way give
    bit-identical losses, coefficients and d_pred to three single-head launches (same blocks, same summation order per head)."""
    B, world, rank = 8, 2, 1
    Ds = [1024, 576 * 64, 24 * 24 * 40]
    g = torch.Generator(device="cuda").manual_seed(12)
    preds = [(torch.randn(B, D, device="cuda", generator=g) * 1.3).to(torch.bfloat16) for D in Ds]
    tgts = [torch.randn(B * world, D, device="cuda", generator=g).to(torch.bfloat16) for D in Ds]
    masks = [torch.ones(B, device="cuda"), torch.tensor([1., 0., 1., 1., 0.5, 1., 1., 1.], device="cuda"), torch.ones(B, device="cuda")]
    scales = [torch.tensor([2.0], device="cuda"), torch.tensor([1.5], device="cuda"), None]
    single = [ops.emb_loss_fwd(p, t, m, s, 0.3, rank=rank) for p, t, m, s in zip(preds, tgts, masks, scales)]
    multi = ops.emb_loss_fwd_multi(preds, tgts, masks, scales, [0.3] * 3, rank=rank)
    for (o1, c1), (o2, c2) in zip(single, multi):
        assert torch.equal(o1, o2) and torch.equal(c1, c2)
    d1 = [ops.emb_loss_bwd(p, t, c, 0.5, rank=rank) for p, t, (_, c) in zip(preds, tgts, single)]
    d2 = ops.emb_loss_bwd_multi(preds, tgts, [c for _, c in multi], [0.5] * 3, rank=rank)
    for a, b in zip(d1, d2):
        assert torch.equal(a, b)
    again = ops.emb_loss_fwd_multi(preds, tgts, masks, scales, [0.3] * 3, rank=rank)          # counters re-armed per task
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(multi, again))
