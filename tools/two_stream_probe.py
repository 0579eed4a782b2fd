"""How much of the persistent GEMMs' XCD tails / dispatch gaps would two half-batch streams through the decoder recover?  (dev tool; gpurun)
Forward chain of L frozen-Llama-3-8B layers on synthetic activations: one stream at M = 16384 rows vs two streams at M = 8192 each, layer launches interleaved
by the single host thread.  Same kernels the engine runs (RoPE epilogue, attention forward, residual GEMMs, fused SwiGLU); no numerics checked here."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from visper_lm_amd import ops
BF = torch.bfloat16
dev = "cuda"
H, I, nh, nkv, hd, S, L = 4096, 14336, 32, 8, 128, 2048, 8
torch.manual_seed(0)
W = [dict(wqkv=(torch.randn((nh + 2 * nkv) * hd, H, device=dev) * 0.02).to(BF), wo=(torch.randn(H, nh * hd, device=dev) * 0.02).to(BF),
          wgu=ops.interleave_gate_up((torch.randn(2 * I, H, device=dev) * 0.02).to(BF)), wd=(torch.randn(H, I, device=dev) * 0.02).to(BF),
          g=torch.ones(H, device=dev, dtype=BF)) for _ in range(L)]
cos_t, sin_t = ops.rope_tables(S, hd, 500000.0, dev)


def layer(x, w, B):
    M = B * S
    xn, _ = ops.rmsnorm_fwd(x, w["g"], 1e-5)
    qkv = ops.gemm_rope(xn, w["wqkv"], S, (nh + nkv) * hd, cos_t, sin_t)
    t3 = qkv.view(B, S, -1)
    q4, k4, v4 = t3[..., :nh * hd].view(B, S, nh, hd), t3[..., nh * hd:(nh + nkv) * hd].view(B, S, nkv, hd), t3[..., (nh + nkv) * hd:].view(B, S, nkv, hd)
    att, _ = ops.attn_fwd(q4, k4, v4, causal=True)
    h1 = ops.gemm(att.view(M, nh * hd), w["wo"], residual=x)
    hn, _ = ops.rmsnorm_fwd(h1, w["g"], 1e-5)
    gu, act = ops.gemm_swiglu_fwd(hn, w["wgu"])
    return ops.gemm(act, w["wd"], residual=h1)


def run_one(B, reps=3):
    x0 = (torch.randn(B * S, H, device=dev) * 0.5).to(BF)
    for _ in range(2):
        x = x0
        for w in W: x = layer(x, w, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        x = x0
        for w in W: x = layer(x, w, B)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run_two(B, reps=3):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    xa0 = (torch.randn(B * S, H, device=dev) * 0.5).to(BF); xb0 = (torch.randn(B * S, H, device=dev) * 0.5).to(BF)
    torch.cuda.synchronize()

    def go():
        xa, xb = xa0, xb0
        for w in W:
            with torch.cuda.stream(s1): xa = layer(xa, w, B)
            with torch.cuda.stream(s2): xb = layer(xb, w, B)
        return xa, xb
    for _ in range(2): go()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


a = run_one(8); b = run_one(4); c = run_two(4)
print(f"{L} layers forward: one stream B=8 {a:.2f} ms | one stream B=4 x 2 sequential {2 * b:.2f} ms | two streams B=4 + B=4 {c:.2f} ms")
