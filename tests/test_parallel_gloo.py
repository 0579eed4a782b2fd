"""world_size-2 gloo test of the two DP exchange points (runs on CPU): rank-ordered target all-gather feeding the
contrastive loss with the `rank*B` label offset (ola_utils.py:104-119), and the split flat-gradient all-reduce."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import visper_oracle as O, weights as WT
        from visper_lm_amd.parallel import all_gather_rows, GradReducer
        import torch.nn.functional as F
        B, D = 3, 40
        pred = WT.tensor(f"p{rank}", (B, 4, 10), 1.2)
        tgt = WT.tensor(f"t{rank}", (B, 4, 10), 1.0)
        gathered = all_gather_rows(tgt.reshape(B, D))
        assert gathered.shape == (world * B, D)
        for r in range(world):                                   # rank order
            assert torch.equal(gathered[r * B:(r + 1) * B], WT.tensor(f"t{r}", (B, 4, 10), 1.0).reshape(B, D))
        e, s1, c = O.emb_loss(pred, torch.ones(B), tgt, torch.tensor(2.0), 0.3, rank=rank, gathered_targets=F.normalize(gathered, dim=-1))
        # single-process restatement: logits against ALL targets, labels offset by rank*B
        p = F.normalize(pred.reshape(B, D), dim=-1)
        allt = F.normalize(torch.cat([WT.tensor(f"t{r}", (B, 4, 10), 1.0).reshape(B, D) for r in range(world)]), dim=-1)
        ce = F.cross_entropy(p @ allt.t() * torch.tensor(2.0).exp(), torch.arange(B) + rank * B)
        assert abs(float(c) - 0.3 * float(ce)) < 1e-6
        # gradient reduction: two pieces, sum then 1/world (mean) in the optimizer
        g = torch.arange(10, dtype=torch.float32) * (rank + 1)
        red = GradReducer(g, split=4)
        red.start_early()
        red.finish()
        assert torch.allclose(g, torch.arange(10, dtype=torch.float32) * sum(r + 1 for r in range(world)))
        # IFT stage: buckets launched in backward order as they become final; finish() reduces exactly the complement, once
        g2 = torch.arange(20, dtype=torch.float32) * (rank + 1)
        red2 = GradReducer(g2, split=0)
        red2.start_early()                                       # empty early block
        red2.reduce_range(12, 20)                                # lm_head + final norm
        red2.reduce_range(8, 12)                                 # last layer
        red2.reduce_range(4, 8)                                  # first layer
        red2.finish()                                            # remaining [0, 4): projector / embeddings
        assert torch.allclose(g2, torch.arange(20, dtype=torch.float32) * sum(r + 1 for r in range(world)))
        assert red2.pending == [] and red2.done == []
        # fp32 on the wire gives the exact sum; the default bf16 buckets (the reference's ZeRO-2 reduces bf16 gradients) round each
        # rank's bucket to bf16, sum, and write the result back into the fp32 buffer
        g3 = (torch.arange(20, dtype=torch.float32) * 0.37 + 0.011) * (rank + 1)
        red3 = GradReducer(g3.clone(), split=8, reduce_dtype=torch.float32)
        red3.start_early(); red3.finish()
        exact = (torch.arange(20, dtype=torch.float32) * 0.37 + 0.011) * sum(r + 1 for r in range(world))
        assert torch.allclose(red3.g, exact, rtol=1e-6)
        red4 = GradReducer(g3.clone(), split=8)
        assert red4.reduce_dtype == torch.bfloat16
        red4.start_early(); red4.finish()
        want = sum(((torch.arange(20, dtype=torch.float32) * 0.37 + 0.011) * (r + 1)).to(torch.bfloat16).float() for r in range(world))
        assert torch.allclose(red4.g, want.to(torch.bfloat16).float(), rtol=1e-2) and torch.allclose(red4.g, exact, rtol=2e-2)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_dp_exchange_points_world2():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))
