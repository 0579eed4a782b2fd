"""Which vp_gemm_bf16 launches of one step differ between the general 4-wave variant and the other kernels?  (dev tool; gpurun)"""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
from visper_lm_amd import ops
from visper_lm_amd.config import llama3_8b
from visper_lm_amd.engine import Engine
dev = torch.device("cuda:0")
ift = len(sys.argv) > 1 and sys.argv[1] == "ift"
cfg = llama3_8b(aux_mode="", num_task_tokens=0, train_llm=True) if ift else llama3_8b()
L = 2
cfg.num_hidden_layers = L
cfg.depth_decoder = not ift
if not ift:
    cfg.image_gen["img_layer_indices"] = "2"; cfg.image_depth["depth_layer_indices"] = "2"; cfg.image_seg["seg_layer_indices"] = "2"
eng = Engine(cfg, device=dev)
eng.set_distributed(0, 1, transport="torch")
eng.init_random(seed=0)
T = 1473 if ift else 1449
b = bench.make_batch(cfg, 8, T, 0, dev)
ids = torch.randint(0, 1000, (8, T)); ids[:, cfg.num_sys_tokens] = -200
lab = ids.clone(); lab[:, :cfg.num_sys_tokens + 7] = -100
b["input_ids"], b["labels"] = ids, lab
orig = ops.gemm
seen = {}
def wrapped(a, w, bias=None, residual=None, epi=0, out=None, out_f32=False, force_generic=False):
    res_copy = residual.clone() if residual is not None else None
    y = orig(a, w, bias=bias, residual=residual, epi=epi, out=out, out_f32=out_f32, force_generic=force_generic)
    if not out_f32 and not force_generic:
        os.environ["X"] = "1"
        M, K = a.reshape(-1, a.shape[-1]).shape; N = w.shape[0]
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        ref = orig(a, w, bias=bias, residual=res_copy, epi=epi, out_f32=False, force_generic=7 if (tiles >= 192 and M >= 256 and N >= 256 and K % 64 == 0) else 2 if K % 64 == 0 else 1)
        key = (M, N, K, bias is not None, residual is not None, epi, tuple(a.stride()), tuple(y.stride()), out is not None and residual is not None and out.data_ptr() == residual.data_ptr())
        d = float((y.float() - ref.float().view_as(y)).abs().max())
        if d > 0 and key not in seen:
            seen[key] = d
            print("DIFF", key, d, flush=True)
    return y
ops.gemm = wrapped
import visper_lm_amd.engine as E
out = eng.train_step(b)
torch.cuda.synchronize()
print("done", float(out["loss"]), len(seen))
