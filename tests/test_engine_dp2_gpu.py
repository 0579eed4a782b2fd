"""End-to-end data-parallel semantics of the ENGINE with two ranks.  On a box with >= 2 devices: one device per rank over RCCL, the step run over
BOTH transports (torch.distributed and the C ABI's vp_comm_*) with bit-identical gradients required.  On the one-GPU test box: both ranks on
device 0, backend gloo (RCCL refuses two ranks on one device).  Each rank runs the tiny PT step on its own half of a 4-sample batch; rank 0
checks its per-layer losses against the fp32 oracle fed the all-gathered targets with the `rank*B` label offset (ola_utils.py:96-125),
and the summed gradients of both ranks against autograd through the oracle on the two half-batches."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, json, torch, torch.distributed as dist
root = os.environ["VP_ROOT"]; sys.path.insert(0, root)
rank = int(os.environ["RANK"]); world = 2
multi = os.environ.get("VP_TEST_MULTI_DEVICE") == "1"     # >= 2 devices on the box: one device per rank over RCCL (both transports)
torch.cuda.set_device(rank if multi else 0)
dist.init_process_group("nccl" if multi else "gloo", rank=rank, world_size=world,
                        **({"device_id": torch.device("cuda", rank)} if multi else {}))
import numpy as np
from oracle import cases, visper_oracle as O, weights as WT
from visper_lm_amd.config import VisperConfig
from visper_lm_amd.engine import Engine
import torch.nn.functional as F
BF = torch.bfloat16
ocfg, W, _, g = cases.tiny_llama_case()
B = 2
def half(r):
    b = cases.make_batch(2 * B, 59, 38)
    return {k: (v[r * B:(r + 1) * B].clone() if torch.is_tensor(v) else v) for k, v in b.items()}
mine = half(rank)
eng = Engine(VisperConfig(**vars(ocfg)))
eng.set_distributed(rank, world)
eng.load_weights(W)
dev = lambda b: {k: (v.cuda() if (k == "images" or k.endswith("_target") or k.endswith("_mask")) else v) for k, v in b.items()}
out = eng.train_step(dev(mine))
eng.finish_grads()                                        # all-reduce (sum) of the flat gradient buffer
torch.cuda.synchronize()
# ---- oracle, single process: both half-batches, targets of BOTH ranks as the contrastive negatives
tr = json.loads(str(g["trainable"]))
Wq = {k: v.to(BF).float() for k, v in W.items()}
for k in tr:
    Wq[k] = Wq[k].clone().requires_grad_(True)
q = lambda b: {k: (v.to(BF).float() if (torch.is_tensor(v) and v.is_floating_point() and not k.endswith("_mask")) else v) for k, v in b.items()}
halves = [q(half(r)) for r in range(world)]
gathered = {t: F.normalize(torch.cat([h[f"{t}_target"].reshape(B, -1) for h in halves]), dim=-1) for t in ("gen", "depth", "seg")}
refs = [O.forward(Wq, halves[r], ocfg, rank=r, gathered=gathered) for r in range(world)]
(refs[0]["loss"] + refs[1]["loss"]).backward()
ref = refs[rank]
sys.path.insert(0, os.path.join(root, "tests"))
from parity import check, rel, grad_err
tag = f"dp2_rank{rank}"
check(f"{tag}/loss_rel", rel(out["loss"], ref["loss"]), 1e-3)
for key, trip in ref["layer_losses"].items():
    got = out["layer_losses"][key].float().cpu().numpy()
    check(f"{tag}/layer_loss/{key[0]}@{key[1]}", max(abs(float(a) - float(b)) / max(abs(float(b)), 1e-6) for a, b in zip(got, trip)), 5e-3)
for k in eng.ps.index:
    got = eng.ps.g(k).detach().float().cpu().reshape(-1)
    want = Wq[k].grad
    if want is None:
        assert float(got.abs().max()) == 0.0, k
        continue
    want = want.reshape(-1)
    if got.numel() == 1:
        check(f"{tag}/grad/{k}_abs", abs(float(got) - float(want)), 0.05 * abs(float(want)) + 1e-3)
        continue
    c, n = grad_err(got, want)
    check(f"{tag}/grad/{k}/one_minus_cos", c, 1.5e-2)
    check(f"{tag}/grad/{k}/norm_dev", n, 3.5e-2)
if multi:
    # the same step over the C ABI's own communicator (vp_comm_*: RCCL + side stream + event fences): bit-identical losses and gradients
    g_torch = eng.ps.grad.detach().clone()
    l_torch = {k: v.clone() for k, v in out["layer_losses"].items()}
    eng.set_distributed(rank, world, transport="native")
    out2 = eng.train_step(dev(mine))
    eng.finish_grads()
    torch.cuda.synchronize()
    assert torch.equal(eng.ps.grad, g_torch), "native transport: all-reduced gradients differ from the torch.distributed leg"
    assert float(out2["loss"]) == float(out["loss"]) and all(torch.equal(out2["layer_losses"][k], v) for k, v in l_torch.items())
    both = [torch.empty_like(g_torch) for _ in range(world)]
    dist.all_gather(both, eng.ps.grad.detach())
    assert torch.equal(both[0], both[1]), "all-reduced gradients differ across ranks"
    print(f"DP2_NATIVE_BITWISE_OK rank {rank}")
dist.barrier()
dist.destroy_process_group()
print(f"DP2_OK rank {rank}")
'''


def test_engine_two_ranks_match_single_process_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    multi = torch.cuda.device_count() >= 2              # then: RCCL, one device per rank, torch AND native transports, success required
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", VP_ROOT=root, RANK=str(r), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", VP_TEST_MULTI_DEVICE="1" if multi else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, o in enumerate(outs):
        assert f"DP2_OK rank {r}" in o, f"rank {r}:\n" + o[-2500:] + "\nother rank:\n" + outs[1 - r][-2500:]
        assert not multi or f"DP2_NATIVE_BITWISE_OK rank {r}" in o, o[-2500:]
