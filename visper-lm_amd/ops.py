"""Tensor-level wrappers over the C ABI (include/visper_hip.h).  PyTorch provides device memory and
streams only; every computation is a libvisper_hip kernel.  All tensors must live on the current HIP
device; bf16 unless stated.  No CPU fallback: a CPU tensor raises."""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import _lib

EPI_NONE, EPI_GELU, EPI_QUICK_GELU, EPI_RELU = 0, 1, 2, 3
BF16 = torch.bfloat16
GEMM_PROF = None   # bench.py sets this to a list: every GEMM launch is bracketed by HIP events on its own stream


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Dynamic tile claims of the persistent 8-phase GEMM kernels (include/visper_hip.h `sched_ws`): the counter blocks are CALLER-owned — one zeroed
# block per (device, stream), allocated here through torch the first time a GEMM is launched on that stream while `set_dynamic(True)`.
_DYNAMIC = [os.environ.get("VP_GEMM_DYN") == "1"]
_SCHED_WS = {}


def set_dynamic(on):
    """Per-XCD dynamic tile claims for the GEMMs launched from here on (Engine.set_distributed: on for world > 1, where RCCL kernels run beside
    the GEMMs).  Returns the previous setting."""
    prev = _DYNAMIC[0]
    _DYNAMIC[0] = bool(on)
    return prev


def _loss_counters():
    """The distillation loss's ticket-counter block for the current stream (include/visper_hip.h `counters`): caller-owned, zeroed once, left
    zeroed by every launch."""
    st = torch.cuda.current_stream()
    key = ("el", st.device.index, st.cuda_stream)
    ws = _SCHED_WS.get(key)
    if ws is None:
        ws = _SCHED_WS[key] = torch.zeros(_lib.raw("vp_emb_loss_counter_bytes") // 4, device=st.device, dtype=torch.int32)
    return C.c_void_p(ws.data_ptr())


def _sched():
    if not _DYNAMIC[0]:
        return None
    st = torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    ws = _SCHED_WS.get(key)
    if ws is None:
        ws = _SCHED_WS[key] = torch.zeros(_lib.raw("vp_gemm_sched_workspace_bytes") // 4, device=st.device, dtype=torch.int32)
    return C.c_void_p(ws.data_ptr())


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("visper_lm_amd ops need device tensors (there is no CPU path)")
    return C.c_void_p(t.data_ptr())


def _rows2d(t):
    """(rows, cols, ld) of a tensor viewed as 2-D with a contiguous last dim."""
    assert t.stride(-1) == 1, "last dim must be contiguous"
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    if t.dim() == 2:
        return t.shape[0], t.shape[1], t.stride(0)
    assert t.is_contiguous(), "3-D+ tensors must be contiguous"
    return t.numel() // t.shape[-1], t.shape[-1], t.shape[-1]


# ------------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, residual=None, epi=EPI_NONE, out=None, out_f32=False, force_generic=False):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias) + residual.
    force_generic: 0 auto | 1 bounds-checked generic kernel | 2 the 128-tile kernel | 3 the simple persistent 256-tile kernel |
    7 the 8-phase kernel | 8 the one-wave-per-SIMD kernel (the auto choice for aligned large problems) | 14 its general variant (bias /
    activation / residual epilogues, M tail)."""
    M, K, lda = _rows2d(a)
    N, K2, ldb = _rows2d(w)
    assert K == K2, (a.shape, w.shape)
    assert a.dtype == BF16 and w.dtype == BF16
    if out is None:
        out = torch.empty(*a.shape[:-1], N, device=a.device, dtype=torch.float32 if out_f32 else BF16)
    _, _, ldc = _rows2d(out)
    ldr = 0
    if residual is not None:
        _, _, ldr = _rows2d(residual)
    if GEMM_PROF is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vp_gemm_bf16", M, N, K, _p(a), lda, _p(w), ldb, _p(out), ldc, _p(bias), _p(residual), ldr, epi,
              1 if out.dtype == torch.float32 else 0, int(force_generic), _sched(), _stream())
    if GEMM_PROF is not None:
        e1.record()
        GEMM_PROF.append((e0, e1, 2.0 * M * N * K, (M, N, K), "nt_lean" if (bias is None and epi == 0) else "nt_epi"))
    return out


def gemm_rowscale_ok(M, N, K):
    """Shapes vp_gemm_bf16_rope accepts as a plain row-scaled GEMM (rope_cols = 0): the general variant of the one-wave-per-SIMD kernel."""
    return N % 256 == 0 and K % 128 == 0 and M >= 256 and ((M + 255) // 256) * (N // 256) >= 64 and os.environ.get("VP_GEMM_ROPE", "1") != "0"


def gemm_rope_ok(M, N, K, head_dim):
    """Shapes vp_gemm_bf16_rope accepts WITH the rotation in its epilogue (whole 128-wide heads per wave sub-tile)."""
    return head_dim == 128 and gemm_rowscale_ok(M, N, K)


def gemm_rope(a, w, S, rope_cols, cos_t, sin_t, pos=None, row_scale=None):
    """qkv = rope(a @ w^T) in one kernel: columns < rope_cols (q and k heads of 128) rotated as rope_() would (bit-identical)."""
    M, K, lda = _rows2d(a)
    N, K2, ldb = _rows2d(w)
    assert K == K2 and a.dtype == BF16 and w.dtype == BF16 and cos_t.dtype == torch.float32 and (rope_cols == 0 or cos_t.shape[-1] == 64)
    out = torch.empty(*a.shape[:-1], N, device=a.device, dtype=BF16)
    if GEMM_PROF is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vp_gemm_bf16_rope", M, N, K, _p(a), lda, _p(w), ldb, _p(out), N, _p(row_scale), rope_cols, _p(cos_t), _p(sin_t), _p(pos), S, _stream())
    if GEMM_PROF is not None:
        e1.record()
        GEMM_PROF.append((e0, e1, 2.0 * M * N * K, (M, N, K), "nt_rope"))
    return out


def fold_norm_ok(M, H, I, head_dim, nqkv):
    """Shapes for which a decoder layer can run with RMSNorm folded away (gamma in the frozen weights, 1/rms as a row scale in the consuming
    GEMM's epilogue, sums of squares from the producing residual GEMM's epilogue): every GEMM involved must be a one-wave-per-SIMD launch."""
    return (gemm_rowscale_ok(M, nqkv, H) and head_dim in (96, 128) and M % 256 == 0 and H % 256 == 0 and I % 128 == 0 and (M // 256) * (2 * I // 256) >= 192
            and os.environ.get("VP_GEMM_W4", "1") == "1" and os.environ.get("VP_FOLD_NORM", "1") != "0")


def gemm_sumsq(a, w, residual):
    """out = a @ w^T + residual and the per-row sum-of-squares partials [M, N/16] of out (vp_gemm_bf16_sumsq)."""
    M, K, lda = _rows2d(a)
    N, K2, ldb = _rows2d(w)
    assert K == K2 and a.dtype == BF16 and w.dtype == BF16
    out = torch.empty(*a.shape[:-1], N, device=a.device, dtype=BF16)
    part = torch.empty(M, N // 16, device=a.device, dtype=torch.float32)
    _, _, ldr = _rows2d(residual)
    if GEMM_PROF is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vp_gemm_bf16_sumsq", M, N, K, _p(a), lda, _p(w), ldb, _p(out), N, _p(residual), ldr, _p(part), _stream())
    if GEMM_PROF is not None:
        e1.record()
        GEMM_PROF.append((e0, e1, 2.0 * M * N * K, (M, N, K), "nt_fold"))
    return out, part


def rstd_from_sumsq(part, H, eps):
    M, nparts = part.shape
    rstd = torch.empty(M, device=part.device, dtype=torch.float32)
    _lib.call("vp_rstd_from_sumsq", M, nparts, _p(part), H, float(eps), _p(rstd), _stream())
    return rstd


def swiglu_fusable(M, N, K):
    """Shapes the fused SwiGLU GEMM epilogues accept (vp_gemm_bf16_swiglu: 8-phase kernel, interior tiles only)."""
    return M % 256 == 0 and N % 256 == 0 and K % 64 == 0


def interleave_gate_up(w_gate_up):
    """[gate; up] row halves (HF gate_proj / up_proj, or Phi-3's fused gate_up_proj) -> rows interleaved in 8-wide chunks
    (g0..7, u0..7, g8..15, ...): the layout of every gate_up / d_gate_up activation in this library."""
    F2 = w_gate_up.shape[0]
    F = F2 // 2
    assert F % 8 == 0
    g, u = w_gate_up[:F], w_gate_up[F:]
    rest = w_gate_up.shape[1:]
    return torch.stack([g.reshape(F // 8, 8, *rest), u.reshape(F // 8, 8, *rest)], 1).reshape(F2, *rest).contiguous()


def gemm_swiglu_fwd(a, w_gu, row_scale=None):
    """gate_up = (row_scale * a) @ w_gu^T (chunk-interleaved), act = silu(gate) * up, in one kernel.  Returns (gate_up, act)."""
    M, K, lda = _rows2d(a)
    N, K2, ldb = _rows2d(w_gu)
    assert K == K2 and swiglu_fusable(M, N, K)
    gu = torch.empty(*a.shape[:-1], N, device=a.device, dtype=BF16)
    act = torch.empty(*a.shape[:-1], N // 2, device=a.device, dtype=BF16)
    if GEMM_PROF is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vp_gemm_bf16_swiglu", 1, M, N, K, _p(a), lda, _p(w_gu), ldb, _p(gu), N, _p(act), N // 2, _p(row_scale), 0, _sched(), _stream())
    if GEMM_PROF is not None:
        e1.record()
        GEMM_PROF.append((e0, e1, 2.0 * M * N * K, (M, N, K), "nt_lean" if row_scale is None else "nt_fold"))
    return gu, act


def gemm_swiglu_bwd(dy, w_down_T, gate_up):
    """d_gate_up = swiglu_bwd(dy @ w_down_T^T, gate_up) in one kernel (d_act never leaves the chip)."""
    M, K, lda = _rows2d(dy)
    N, K2, ldb = _rows2d(w_down_T)
    assert K == K2 and swiglu_fusable(M, N, K) and gate_up.shape[-1] == 2 * N and gate_up.is_contiguous()
    dgu = torch.empty_like(gate_up)
    if GEMM_PROF is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vp_gemm_bf16_swiglu", 2, M, N, K, _p(dy), lda, _p(w_down_T), ldb, _p(dgu), 2 * N, None, 0, _p(gate_up), 2 * N,
              _sched(), _stream())
    if GEMM_PROF is not None:
        e1.record()
        GEMM_PROF.append((e0, e1, 2.0 * M * N * K, (M, N, K), "nt_lean"))
    return dgu


def gemm_tn_ok(M, N, K):
    """Shapes the TN (transpose-free weight-gradient) kernel takes."""
    return M % 256 == 0 and N % 256 == 0 and K % 64 == 0


def gemm_tn(a, b, out=None, out_f32=True, accumulate=False):
    """out[M,N] (+)= a[K,M]^T @ b[K,N]  (dW = dY^T X from row-major activations; no transposed copies)."""
    K, M, lda = _rows2d(a)
    K2, N, ldb = _rows2d(b)
    assert K == K2 and gemm_tn_ok(M, N, K)
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else BF16)
    assert out.stride(-1) == 1 and out.dtype == (torch.float32 if out_f32 else BF16)
    if GEMM_PROF is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vp_gemm_tn_bf16", M, N, K, _p(a), lda, _p(b), ldb, _p(out), out.stride(0), int(out_f32), int(accumulate), _sched(), _stream())
    if GEMM_PROF is not None:
        e1.record()
        GEMM_PROF.append((e0, e1, 2.0 * M * N * K, (M, N, K), "tn"))
    return out


def transpose(x, out=None):
    """2-D transpose (bf16)."""
    R, Cc, ldi = _rows2d(x)
    if out is None:
        out = torch.empty(Cc, R, device=x.device, dtype=BF16)
    _lib.call("vp_transpose_bf16", R, Cc, _p(x), ldi, _p(out), out.stride(0), _stream())
    return out


def transpose_batched(x, out=None):
    """[B, R, C] -> [B, C, R] (bf16, contiguous), one launch."""
    B, R, Cc = x.shape
    assert x.is_contiguous() and x.dtype == BF16
    if out is None:
        out = torch.empty(B, Cc, R, device=x.device, dtype=BF16)
    _lib.call("vp_transpose_batched_bf16", B, R, Cc, _p(x), R * Cc, Cc, _p(out), R * Cc, R, _stream())
    return out


def rmsnorm_fwd(x, w, eps, save_rstd=True):
    M, H, ldx = _rows2d(x)
    y = torch.empty(x.shape, device=x.device, dtype=BF16)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if save_rstd else None
    _lib.call("vp_rmsnorm_fwd", M, H, _p(x), ldx, _p(w), eps, _p(y), H, _p(rstd), _stream())
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dres=None):
    M, H, ld = _rows2d(x)
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    _lib.call("vp_rmsnorm_bwd", M, H, _p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), ld, _stream())
    return dx


def layernorm_fwd(x, w, b, eps=1e-5, save_stats=True):
    M, H, ldx = _rows2d(x)
    y = torch.empty(x.shape, device=x.device, dtype=BF16)
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if save_stats else None
    _lib.call("vp_layernorm_fwd", M, H, _p(x), ldx, _p(w), _p(b), eps, _p(y), H, _p(mean), _p(rstd), _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dres=None, want_wb=True, dw_out=None, db_out=None):
    """-> dx (bf16), dw (f32), db (f32); dw_out / db_out: fp32 [H] destinations (e.g. views of the flat gradient buffer) written in place."""
    M, H, ld = _rows2d(x)
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    _lib.call("vp_layernorm_bwd_dx", M, H, _p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), ld, _stream())
    if not want_wb:
        return dx, None, None
    rpb = max(1, (M + 255) // 256)
    nslab = (M + rpb - 1) // rpb
    pw = torch.empty(nslab, H, device=x.device, dtype=torch.float32)
    pb = torch.empty(nslab, H, device=x.device, dtype=torch.float32)
    _lib.call("vp_layernorm_bwd_wb_partial", M, H, _p(dy), _p(x), _p(mean), _p(rstd), _p(pw), _p(pb), ld, rpb, _stream())
    dw = torch.empty(H, device=x.device, dtype=torch.float32) if dw_out is None else dw_out
    db = torch.empty(H, device=x.device, dtype=torch.float32) if db_out is None else db_out
    _lib.call("vp_colsum_finish", nslab, H, _p(pw), _p(dw), 1.0, 0, _stream())
    _lib.call("vp_colsum_finish", nslab, H, _p(pb), _p(db), 1.0, 0, _stream())
    return dx, dw, db


_ZEROS_F32 = {}


def rmsnorm_bwd_w(dy, x, rstd, out=None):
    """RMSNorm weight gradient sum_rows dy * x * rstd -> f32 [H] (the LayerNorm partial kernel with mean = 0)."""
    M, H, ld = _rows2d(x)
    assert dy.is_contiguous() and x.is_contiguous()
    rpb = max(1, (M + 255) // 256)
    nslab = (M + rpb - 1) // rpb
    pw = torch.empty(nslab, H, device=x.device, dtype=torch.float32)
    pb = torch.empty(nslab, H, device=x.device, dtype=torch.float32)
    zero = _ZEROS_F32.get(x.device)                     # ONE zeros buffer per device, grown to the largest M seen (ragged batches change M every step)
    if zero is None or zero.numel() < M:
        zero = _ZEROS_F32[x.device] = torch.zeros(max(M, 2 * (zero.numel() if zero is not None else 0)), device=x.device, dtype=torch.float32)
    _lib.call("vp_layernorm_bwd_wb_partial", M, H, _p(dy), _p(x), _p(zero), _p(rstd), _p(pw), _p(pb), ld, rpb, _stream())
    dw = torch.empty(H, device=x.device, dtype=torch.float32) if out is None else out
    _lib.call("vp_colsum_finish", nslab, H, _p(pw), _p(dw), 1.0, 0, _stream())
    return dw


def scatter_add_rows_(dst_f32, src, idx):
    """dst[idx[r]] += src[r] (fp32 atomics; idx < 0 skips)."""
    n, H, lds = _rows2d(src)
    assert dst_f32.dtype == torch.float32 and dst_f32.is_contiguous() and idx.dtype == torch.int32
    _lib.call("vp_scatter_add_rows", n, H, _p(src), lds, _p(idx), _p(dst_f32), _stream())
    return dst_f32


def rope_tables(S, head_dim, theta, device):
    """cos/sin [S, head_dim/2] fp32, rounded to bf16 first (HF casts the tables to the activation dtype)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
    return (fr.cos().to(BF16).float().to(device).contiguous(), fr.sin().to(BF16).float().to(device).contiguous())


def rope_(x2d, T, S, nheads, head_dim, cos_t, sin_t, pos=None, inverse=False):
    """In-place rotate-half RoPE on a [T, >= nheads*head_dim] row-strided slice."""
    _lib.call("vp_rope", T, S, nheads, head_dim, _p(x2d), x2d.stride(0), _p(cos_t), _p(sin_t), _p(pos), 1 if inverse else 0,
              _stream())
    return x2d


def dwconv7x7_nhwc(x, w_tap_major, bias):
    """x [B,H,W,C] bf16 channels-last, w [49,C], bias [C] -> y [B,H,W,C] (zero padding 3)."""
    B, Hh, Ww, Cc = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.call("vp_dwconv7x7_nhwc", B, Hh, Ww, Cc, _p(x), _p(w_tap_major), _p(bias), _p(y), _stream())
    return y


# ---- NHWC helpers for the frozen DPT depth decoder (conv.hip)
def conv3x3_nhwc(x, w_mat, bias=None, stride=1, relu_in=False, epi=EPI_NONE, residual=None):
    """x: [B,H,W,C] bf16; w_mat: [Cout, 9*C] with column order (ky, kx, c); pad 1.  -> [B,Ho,Wo,Cout]."""
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    col = torch.empty(B * Ho * Wo, 9 * C, device=x.device, dtype=BF16)
    _lib.call("vp_im2col3x3_nhwc", B, H, W, C, stride, 1 if relu_in else 0, _p(x), _p(col), _stream())
    res2 = None if residual is None else residual.reshape(B * Ho * Wo, -1)
    return gemm(col, w_mat, bias=bias, epi=epi, residual=res2).view(B, Ho, Wo, -1)


def bilinear_nhwc(x, Ho, Wo, align_corners=True):
    """F.interpolate(mode="bilinear", align_corners=...) on [B,H,W,C] bf16."""
    B, H, W, C = x.shape
    y = torch.empty(B, Ho, Wo, C, device=x.device, dtype=BF16)
    _lib.call("vp_bilinear_nhwc", B, H, W, C, Ho, Wo, 1 if align_corners else 0, _p(x), _p(y), _stream())
    return y


def conv_transpose_nhwc(x, w_mat, bias_rep, k):
    """ConvTranspose2d(kernel = stride = k): w_mat [k*k*Cout, Cin] (rows (ky, kx, co)), bias_rep = bias tiled k*k times."""
    B, H, W, C = x.shape
    t = gemm(x.reshape(B * H * W, C), w_mat, bias=bias_rep)
    Co = w_mat.shape[0] // (k * k)
    y = torch.empty(B, H * k, W * k, Co, device=x.device, dtype=BF16)
    _lib.call("vp_pixel_shuffle_nhwc", B, H, W, k, Co, _p(t), _p(y), _stream())
    return y


def minmax_norm(x):
    """Per leading-index (x - min) / (max - min), bf16 rounding after each op."""
    B = x.shape[0]
    y = torch.empty_like(x)
    _lib.call("vp_minmax_norm", B, x[0].numel(), _p(x), _p(y), _stream())
    return y


def swiglu_fwd(gate_up, interleaved=True):
    """gate_up: [..., 2F], chunk-interleaved (see interleave_gate_up) or [gate | up] halves -> silu(gate) * up [..., F]."""
    M, F2, ldg = _rows2d(gate_up)
    F = F2 // 2
    out = torch.empty(*gate_up.shape[:-1], F, device=gate_up.device, dtype=BF16)
    _lib.call("vp_swiglu_fwd", M, F, _p(gate_up), ldg, _p(out), F, 1 if interleaved else 0, _stream())
    return out


def swiglu_bwd(dact, gate_up, interleaved=True):
    M, F2, ldg = _rows2d(gate_up)
    F = F2 // 2
    _, _, ldd = _rows2d(dact)
    dgu = torch.empty_like(gate_up)
    _lib.call("vp_swiglu_bwd", M, F, _p(dact), ldd, _p(gate_up), _p(dgu), ldg, 1 if interleaved else 0, _stream())
    return dgu


def act_fwd(x, kind):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.call("vp_act_fwd", kind, x.numel(), _p(x), _p(y), _stream())
    return y


def act_bwd(dy, x, kind):
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    _lib.call("vp_act_bwd", kind, x.numel(), _p(dy), _p(x), _p(dx), _stream())
    return dx


def add(a, b, out=None):
    assert a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    _lib.call("vp_add_bf16", a.numel(), _p(a), _p(b), _p(out), _stream())
    return out


def add2d_(dst, src):
    R, Cc, ldd = _rows2d(dst)
    _, _, lds = _rows2d(src)
    _lib.call("vp_add2d_bf16", R, Cc, _p(dst), ldd, _p(src), lds, _stream())
    return dst


def zero_(t):
    """t[...] = 0 for a contiguous tensor (hipMemsetAsync on the current stream)."""
    assert t.is_contiguous()
    _lib.call("vp_memset_zero", _p(t), t.numel() * t.element_size(), _stream())
    return t


def copy2d_(dst, src):
    R, Cc, ldd = _rows2d(dst)
    _, _, lds = _rows2d(src)
    _lib.call("vp_copy2d_bf16", R, Cc, _p(dst), ldd, _p(src), lds, _stream())
    return dst


def colsum(x, out=None, accumulate=False, scale=1.0):
    """fp32 column sums of a bf16 [M,N] matrix (bias gradient)."""
    M, N, ld = _rows2d(x)
    rpb = max(1, (M + 255) // 256)
    nslab = (M + rpb - 1) // rpb
    part = torch.empty(nslab, N, device=x.device, dtype=torch.float32)
    _lib.call("vp_colsum_partial", M, N, _p(x), ld, _p(part), rpb, _stream())
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
        accumulate = False
    _lib.call("vp_colsum_finish", nslab, N, _p(part), _p(out), scale, 1 if accumulate else 0, _stream())
    return out


def gather_rows(srcs, kind, row, H, out):
    """out[i] = srcs[kind[i]][row[i]] (kind<0 -> zeros).  srcs: list of <=4 2-D bf16 tensors; kind,row int32."""
    n = out.numel() // H
    arr_p = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
    arr_l = (C.c_long * len(srcs))(*[_rows2d(t)[2] for t in srcs])
    _lib.call("vp_gather_rows", n, H, arr_p, arr_l, len(srcs), _p(kind), _p(row), _p(out), H, _stream())
    return out


def gather_sum_rows(src, idx, cnt, scale, out, accumulate=False):
    """out[i] = scale * sum_k src[idx[i*cnt+k]] (idx<0 skipped). src bf16/f32 2-D, out bf16/f32 2-D."""
    n, H, ldo = _rows2d(out)
    _, _, lds = _rows2d(src)
    _lib.call("vp_gather_sum_rows", n, cnt, H, _p(src), lds, 1 if src.dtype == torch.float32 else 0, _p(idx), scale, _p(out), ldo,
              1 if out.dtype == torch.float32 else 0, 1 if accumulate else 0, _stream())
    return out


def cast_to_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=BF16)
    _lib.call("vp_cast_f32_to_bf16", x.numel(), _p(x), _p(out), _stream())
    return out


def cast_to_f32(x, out=None, accumulate=False):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        accumulate = False
    _lib.call("vp_cast_bf16_to_f32", x.numel(), _p(x), _p(out), 1 if accumulate else 0, _stream())
    return out


def scatter_rows_to_f32(src, idx, dst):
    """dst[idx[r]] = float(src[r]) (idx None: row r; idx < 0 skips): bf16 rows widened into rows of an fp32 2-D tensor."""
    n, H, lds = _rows2d(src)
    _, H2, ldd = _rows2d(dst)
    assert H == H2 and src.dtype == BF16 and dst.dtype == torch.float32 and (idx is None or idx.dtype == torch.int32)
    _lib.call("vp_scatter_rows_bf16_to_f32", n, H, _p(src), lds, _p(idx), _p(dst), ldd, _stream())
    return dst


def sumsq(x):
    """sum(x^2) of an fp32 tensor -> fp32 [1] (deterministic two-stage reduction)."""
    n = x.numel()
    part = torch.empty(_lib.raw("vp_sumsq_nblk", n), device=x.device, dtype=torch.float32)
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    _lib.call("vp_sumsq_f32", n, _p(x), _p(part), _p(out), _stream())
    return out


def sum_f32(x, scale=1.0, out=None):
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
    _lib.call("vp_sum_f32", x.numel(), _p(x), _p(out), scale, _stream())
    return out


# ------------------------------------------------------------------------------------------------
def _bshd(t):
    """strides (batch, token) of a [B,S,H,D] view whose (H,D) block is contiguous."""
    B, S, H, D = t.shape
    assert t.stride(3) == 1 and t.stride(2) == D, "heads must be contiguous within a token row"
    return t.stride(0), t.stride(1)


def attn_fwd(q, k, v, causal, scale=None, window=0, kv_len=None, out=None):
    """q [B,Sq,Hq,D], k/v [B,Skv,Hkv,D] (views allowed) -> o [B,Sq,Hq,D], lse2 [B,Hq,Sq] (log2 domain)."""
    B, Sq, Hq, D = q.shape
    _, Skv, Hkv, _ = k.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if out is None:
        out = torch.empty(B, Sq, Hq, D, device=q.device, dtype=BF16)
    lse = torch.empty(B, Hq, Sq, device=q.device, dtype=torch.float32)
    qb, qt = _bshd(q); kb, kt = _bshd(k); vb, vt = _bshd(v); ob, ot = _bshd(out)
    _lib.call("vp_attn_fwd", B, Hq, Hkv, Sq, Skv, D, _p(q), qb, qt, _p(k), kb, kt, _p(v), vb, vt, _p(out), ob, ot, _p(lse),
              _p(kv_len), 1 if causal else 0, window, scale, _stream())
    return out, lse


def attn_fwd_bias(q, k, v, bias_h=None, bias_b=None, scale=None):
    """Non-causal attention with additive fp32 score biases (Swin window attention): bias_h [Hq,Sq,Skv] per head, bias_b [nb,Sq,Skv]
    indexed by batch % nb.  Forward only.  -> o [B,Sq,Hq,D]."""
    B, Sq, Hq, D = q.shape
    _, Skv, Hkv, _ = k.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    out = torch.empty(B, Sq, Hq, D, device=q.device, dtype=BF16)
    lse = torch.empty(B, Hq, Sq, device=q.device, dtype=torch.float32)
    for t in (bias_h, bias_b):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.shape[-2:] == (Sq, Skv))
    qb, qt = _bshd(q); kb, kt = _bshd(k); vb, vt = _bshd(v); ob, ot = _bshd(out)
    _lib.call("vp_attn_fwd_bias", B, Hq, Hkv, Sq, Skv, D, _p(q), qb, qt, _p(k), kb, kt, _p(v), vb, vt, _p(out), ob, ot, _p(lse),
              None, 0, 0, scale, _p(bias_h), _p(bias_b), 0 if bias_b is None else bias_b.shape[0], _stream())
    return out


def attn_bwd(q, k, v, o, lse, dout, causal, scale=None, window=0, kv_len=None, dq=None, dk=None, dv=None, rope=None):
    B, Sq, Hq, D = q.shape
    _, Skv, Hkv, _ = k.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if dq is None:
        dq = torch.empty(B, Sq, Hq, D, device=q.device, dtype=BF16)
    if dk is None:
        dk = torch.empty(B, Skv, Hkv, D, device=q.device, dtype=BF16)
    if dv is None:
        dv = torch.empty(B, Skv, Hkv, D, device=q.device, dtype=BF16)
    delta = torch.empty(3, B, Hq, Sq, device=q.device, dtype=torch.float32)   # delta + (lse, delta) pairs
    qb, qt = _bshd(q); kb, kt = _bshd(k); vb, vt = _bshd(v); ob, ot = _bshd(o); gb, gt = _bshd(dout)
    dqb, dqt = _bshd(dq); dkb, dkt = _bshd(dk); dvb, dvt = _bshd(dv)
    if rope is not None:                                  # (cos, sin[, pos]): dq / dk come out rotated back (D = 128 / 96, causal)
        cos_t, sin_t = rope[0], rope[1]
        pos = rope[2] if len(rope) > 2 else None
        _lib.call("vp_attn_bwd_rope", B, Hq, Hkv, Sq, Skv, D, _p(q), qb, qt, _p(k), kb, kt, _p(v), vb, vt, _p(o), ob, ot, _p(lse),
                  _p(dout), gb, gt, _p(dq), dqb, dqt, _p(dk), dkb, dkt, _p(dv), dvb, dvt, _p(delta), _p(kv_len),
                  1 if causal else 0, window, scale, _p(cos_t), _p(sin_t), _p(pos), _stream())
        return dq, dk, dv
    _lib.call("vp_attn_bwd", B, Hq, Hkv, Sq, Skv, D, _p(q), qb, qt, _p(k), kb, kt, _p(v), vb, vt, _p(o), ob, ot, _p(lse),
              _p(dout), gb, gt, _p(dq), dqb, dqt, _p(dk), dkb, dkt, _p(dv), dvb, dvt, _p(delta), _p(kv_len),
              1 if causal else 0, window, scale, _stream())
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------
def ce_fwd_bwd(logits, labels, grad_scale, write_grad=True, out=None):
    """logits [rows, V] bf16 (overwritten with dlogits when write_grad), labels int64 [rows] -> row_loss f32 [rows] (`out`: destination)."""
    rows, V, ld = _rows2d(logits)
    row_loss = torch.empty(rows, device=logits.device, dtype=torch.float32) if out is None else out
    _lib.call("vp_ce_fwd_bwd", rows, V, _p(logits), ld, _p(labels), _p(row_loss), grad_scale, 1 if write_grad else 0, _stream())
    return row_loss


def emb_loss_fwd(pred, tgt_all, mask, logit_scale, w_con, rank=0):
    """pred [B, D] bf16, tgt_all [Bw, D] bf16 (rank-ordered gather), mask f32 [B], logit_scale f32 [1] or None.
    -> out3 f32 [3] = (emb_loss, sl1, contrastive), coef (saved for backward)."""
    B, D = pred.shape
    Bw = tgt_all.shape[0]
    assert pred.is_contiguous() and tgt_all.is_contiguous() and tgt_all.shape[1] == D
    part = torch.empty(max(1, _lib.raw("vp_emb_loss_workspace", B, Bw, D)), device=pred.device, dtype=torch.float32)
    coef = torch.empty(2 * B + B * Bw + 1, device=pred.device, dtype=torch.float32)
    out3 = torch.empty(3, device=pred.device, dtype=torch.float32)
    _lib.call("vp_emb_loss_fwd", B, Bw, D, rank, _p(pred), _p(tgt_all), _p(mask), _p(logit_scale), w_con, _p(out3), _p(coef),
              _p(part), _loss_counters(), _stream())
    return out3, coef


def emb_loss_bwd(pred, tgt_all, coef, grad_out, rank=0):
    B, D = pred.shape
    dpred = torch.empty_like(pred)
    _lib.call("vp_emb_loss_bwd", B, tgt_all.shape[0], D, rank, _p(pred), _p(tgt_all), _p(coef), grad_out, _p(dpred), _stream())
    return dpred


def emb_loss_fwd_multi(preds, tgts, masks, scales, w_cons, rank=0):
    """Every distillation head of the step in ONE launch (vp_emb_loss_fwd_multi): preds[t] [B, D_t] bf16, tgts[t] [Bw, D_t] bf16, masks[t] f32 [B],
    scales[t] f32 [1] or None, w_cons[t] float.  -> [(out3, coef)] per head."""
    n = len(preds)
    B, Bw = preds[0].shape[0], tgts[0].shape[0]
    dev = preds[0].device
    Ds = (C.c_long * n)(*[int(x.shape[1]) for x in preds])
    outs = []
    parts = []
    for t in range(n):
        assert preds[t].is_contiguous() and tgts[t].is_contiguous() and preds[t].shape[0] == B and tgts[t].shape == (Bw, preds[t].shape[1])
        parts.append(torch.empty(max(1, _lib.raw("vp_emb_loss_workspace", B, Bw, int(preds[t].shape[1]))), device=dev, dtype=torch.float32))
        outs.append((torch.empty(3, device=dev, dtype=torch.float32), torch.empty(2 * B + B * Bw + 1, device=dev, dtype=torch.float32)))
    arr = lambda xs: (C.c_void_p * n)(*[None if x is None else x.data_ptr() for x in xs])
    wc = (C.c_float * n)(*[float(w) for w in w_cons])
    _lib.call("vp_emb_loss_fwd_multi", n, B, Bw, Ds, rank, arr(preds), arr(tgts), arr(masks), arr(scales), wc, arr([o[0] for o in outs]),
              arr([o[1] for o in outs]), arr(parts), _loss_counters(), _stream())
    return outs


def emb_loss_bwd_multi(preds, tgts, coefs, grad_outs, rank=0):
    n = len(preds)
    B, Bw = preds[0].shape[0], tgts[0].shape[0]
    Ds = (C.c_long * n)(*[int(x.shape[1]) for x in preds])
    dpreds = [torch.empty_like(x) for x in preds]
    arr = lambda xs: (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    go = (C.c_float * n)(*[float(g) for g in grad_outs])
    _lib.call("vp_emb_loss_bwd_multi", n, B, Bw, Ds, rank, arr(preds), arr(tgts), arr(coefs), go, arr(dpreds), _stream())
    return dpreds


def adamw_(p, g, m, v, shadow, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    _lib.call("vp_adamw", p.numel(), _p(p), _p(g), _p(m), _p(v), _p(shadow), lr, beta1, beta2, eps, wd, step, grad_scale, _stream())
