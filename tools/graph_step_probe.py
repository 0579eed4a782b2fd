"""Does a captured HIP graph of the whole PT step shrink the kernel-to-kernel gaps?  (dev tool; run via gpurun)
Same engine / batch construction as bench.py (configs[1]); times K eager steps, then K replays of ONE captured step (train_step + optimizer)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from visper_lm_amd.config import llama3_8b
from visper_lm_amd.engine import Engine

dev = torch.device("cuda:0")
cfg = llama3_8b()
cfg.depth_decoder = True
if len(sys.argv) > 1:
    L = int(sys.argv[1])
    cfg.num_hidden_layers = L
    cfg.image_gen["img_layer_indices"] = str(min(20, L)); cfg.image_depth["depth_layer_indices"] = str(min(18, L)); cfg.image_seg["seg_layer_indices"] = str(min(18, L))
eng = Engine(cfg, device=dev)
eng.set_distributed(0, 1, transport="torch")
eng.init_random(seed=0)
b = bench.make_batch(cfg, 8, 1449, 0, dev)
ids = torch.randint(0, 1000, (8, 1449)); ids[:, cfg.num_sys_tokens] = -200
lab = ids.clone(); lab[:, :cfg.num_sys_tokens + 7] = -100
b["input_ids"], b["labels"] = ids, lab

def step():
    out = eng.train_step(b)
    eng.optimizer_step(lr=1e-3, lr_mult=1.0)
    return out

for _ in range(3): step()
torch.cuda.synchronize()
K = 8
t0 = time.perf_counter()
for _ in range(K): out = step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / K * 1e3:.2f} ms/step, loss {float(out['loss']):.4f}", flush=True)
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): g.replay()
    torch.cuda.synchronize()
    print(f"graph: {(time.perf_counter() - t0) / K * 1e3:.2f} ms/step, loss {float(out['loss']):.4f}", flush=True)
except Exception as e:
    import traceback; traceback.print_exc()
    print("capture failed:", repr(e)[:400])
