"""Attention micro-benchmark at the Llama-3-8B train-step shape (dev tool; run through gpurun)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops

B, Hq, Hkv, S, D = 8, 32, 8, 2048, 128
qkv = torch.randn(B, S, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * D].unflatten(-1, (Hq, D)); k = qkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); v = qkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))
do = torch.randn(B, S, Hq, D, device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq = dqkv[..., :Hq * D].unflatten(-1, (Hq, D)); dk = dqkv[..., Hq * D:(Hq + Hkv) * D].unflatten(-1, (Hkv, D)); dv = dqkv[..., (Hq + Hkv) * D:].unflatten(-1, (Hkv, D))

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

o, lse = ops.attn_fwd(q, k, v, True)
fl = 4.0 * B * Hq * S * S * D / 2           # causal fwd flops
ms = t(lambda: ops.attn_fwd(q, k, v, True))
print(f"fwd  {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
ms = t(lambda: ops.attn_bwd(q, k, v, o, lse, do, True, dq=dq, dk=dk, dv=dv))
print(f"bwd  {ms:.3f} ms  {2.5 * fl / ms / 1e9:.0f} TF/s (algorithmic 2.5x fwd)")

# comparison baseline only (like hipBLASLt in tools/gemm_bench.py): PyTorch's scaled_dot_product_attention (AOTriton flash kernels on ROCm), GQA
# expanded to 32 kv heads, same causal shape; forward and forward+backward
try:
    import torch.nn.functional as F
    qh = q.permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    kh = k.permute(0, 2, 1, 3).repeat_interleave(Hq // Hkv, 1).contiguous().requires_grad_(True)
    vh = v.permute(0, 2, 1, 3).repeat_interleave(Hq // Hkv, 1).contiguous().requires_grad_(True)
    doh = do.permute(0, 2, 1, 3).contiguous()
    from torch.nn.attention import sdpa_kernel, SDPBackend
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        ms = t(lambda: F.scaled_dot_product_attention(qh, kh, vh, is_causal=True), n=10)
        print(f"torch SDPA (flash / AOTriton) fwd  {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
        oh = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True)
        def fb():
            oh_ = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True)
            oh_.backward(doh)
        msb = t(fb, n=10)
        print(f"torch SDPA fwd+bwd {msb:.3f} ms -> bwd ~{msb - ms:.3f} ms  {2.5 * fl / (msb - ms) / 1e9:.0f} TF/s")
except Exception as e:  # noqa: BLE001
    print("torch SDPA comparison unavailable:", type(e).__name__, str(e)[:200])
