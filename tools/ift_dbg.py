import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
from visper_lm_amd.config import llama3_8b
from visper_lm_amd.engine import Engine
dev = torch.device("cuda:0")
cfg = llama3_8b(aux_mode="", num_task_tokens=0, train_llm=True)
cfg.num_hidden_layers = 4
cfg.depth_decoder = False
eng = Engine(cfg, device=dev)
eng.set_distributed(0, 1, transport="torch")
eng.init_random(seed=0)
pool = [bench.make_batch(cfg, 8, 1473, 1000 * j, dev) for j in range(4)]
torch.cuda.synchronize()
from visper_lm_amd import ops as _ops
trace = []
def _wrap(name):
    f = getattr(_ops, name)
    def g(*a, **k):
        y = f(*a, **k)
        if torch.cuda.current_stream() != torch.cuda.default_stream():
            t = y[0] if isinstance(y, tuple) else y
            ins = [x for x in a if torch.is_tensor(x) and x.is_floating_point()]
            trace.append((name, tuple(t.shape), torch.isnan(t.float()).any(), [torch.isnan(x.float()).any() for x in ins], k.get("epi", 0), k.get("bias") is not None, k.get("residual") is not None))
        return y
    setattr(_ops, name, g)
for n in ("gemm", "layernorm_fwd", "attn_fwd", "act_fwd", "add"):
    if hasattr(_ops, n): _wrap(n)
flags = []
_vf = eng.vit_forward
def vf(images):
    y = _vf(images)
    eng._last_feats = y
    return y
eng.vit_forward = vf
gi = torch.Generator().manual_seed(4321)
for it in range(6):
    b = dict(pool[it % 4])
    ids = torch.randint(0, 1000, (8, 1473), generator=gi); ids[:, cfg.num_sys_tokens] = -200
    lab = ids.clone(); lab[:, :cfg.num_sys_tokens + 7] = -100
    b["input_ids"], b["labels"] = ids, lab
    b["images_resident"] = True
    out = eng.train_step(b)
    if it == 1:
        keep = (eng._last_feats, out["image_features"])
        tr1 = list(trace)
    trace.clear()
    flags.append((out["loss"].clone(), torch.isnan(out["image_features"].float()).any(), out["image_features"].float().abs().max(),
                  torch.isnan(out["inputs_embeds"].float()).any(), torch.isnan(out["hidden"].float()).any(), torch.isnan(eng.ps.grad).any()))
    eng.optimizer_step(lr=1e-3, lr_mult=1.0)
    if os.environ.get("SYNC"): torch.cuda.synchronize()
torch.cuda.synchronize()
for it, f in enumerate(flags):
    print(it, "loss", float(f[0]), "img nan", bool(f[1]), "img max", float(f[2]), "x nan", bool(f[3]), "hidden nan", bool(f[4]), "grad nan", bool(f[5]), flush=True)

f, img = keep
fn = torch.isnan(f.float())
print("feats", tuple(f.shape), "nan rows", int(fn.any(1).sum()), "nan cols", int(fn.any(0).sum()), "first nan rows", fn.any(1).nonzero().flatten()[:12].tolist(), "first nan cols", fn.any(0).nonzero().flatten()[:12].tolist())
rn = fn.any(1).nonzero().flatten()
if rn.numel():
    import collections
    print("nan row blocks of 256:", sorted(collections.Counter((rn // 256).tolist()).items())[:30])
    cn = fn.any(0).nonzero().flatten()
    print("nan col blocks of 128:", sorted(collections.Counter((cn // 128).tolist()).items())[:40])
bad = [k for k, v in eng.fz.items() if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v.float()).all()]
print("non-finite frozen tensors:", bad[:20], len(bad))
bad2 = [str(k) for k, v in eng._static.items() if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v.float()).all()]
print("non-finite static tables:", bad2[:10])
print("images finite:", [bool(torch.isfinite(p["images"].float()).all()) for p in pool])
torch.cuda.synchronize()
y = _vf(pool[1]["images"])
print("tower alone after the run: nan", bool(torch.isnan(y.float()).any()))

first = None
for j, (name, shp, fl, insf, epi, hb, hr) in enumerate(tr1):
    if bool(fl) and first is None:
        first = j
        print("FIRST NaN output: op", j, name, shp, "inputs nan:", [bool(x) for x in insf], "epi", epi, "bias", hb, "res", hr)
        for jj in range(max(0, j - 3), j):
            print("   before:", jj, tr1[jj][0], tr1[jj][1], bool(tr1[jj][2]))
print("ops traced", len(tr1))
