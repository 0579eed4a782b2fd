"""Data-parallel exchange points of the PT step over torch.distributed — backend "nccl" (= RCCL over
xGMI) on MI355X, "gloo" in the CPU tests.  One process per GPU; exactly two collectives per step:

  1. all_gather_rows: the frozen-teacher target features, once per task per step (the reference re-gathers the
     normalised targets in every layer call: ola_utils.py:96-106,118-119) -> rank-ordered [world*B, D].
  2. GradReducer: sum-all-reduce of the flat fp32 gradient buffer in two pieces — heads + logit scales as soon as the
     heads' backward is done (their all-reduce runs on RCCL's stream underneath the whole decoder backward), projector +
     task tokens at the end.  The 1/world mean is folded into the fused AdamW (grad_scale).  Replaces DeepSpeed ZeRO-2's
     bucketed reduce-scatter (scripts/zero2.json:16-22, overlap_comm:false).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_rows(flat: torch.Tensor) -> torch.Tensor:
    """[B, D] on every rank -> [world*B, D], rank r's rows at r*B..(r+1)*B (dist_collect order, ola_utils.py:104-106)."""
    rank, world = world_info()
    if world == 1:
        return flat
    if dist.get_backend() == "gloo":                   # CPU test backend: no bf16 / device all_gather_into_tensor -> stage via fp32 host
        parts = [torch.empty(flat.shape, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, flat.detach().float().cpu().contiguous())
        return torch.cat(parts, 0).to(device=flat.device, dtype=flat.dtype)
    out = torch.empty(world * flat.shape[0], flat.shape[1], device=flat.device, dtype=flat.dtype)
    dist.all_gather_into_tensor(out, flat.contiguous())
    return out


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, split: int):
        self.g, self.split, self.pending = flat_grad, split, []
        self.done = []                                  # [lo, hi) ranges already launched this step

    def _launch(self, lo, hi):
        _, world = world_info()
        if hi > lo:
            self.done.append((lo, hi))
            if world > 1:
                self.pending.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def start_early(self):
        """heads + logit-scale gradients are final: reduce them while the decoder backward runs."""
        self._launch(0, self.split)

    def reduce_range(self, lo, hi):
        """IFT stage: one bucket (lm_head + final norm, then one decoder layer at a time, in backward order) whose gradients are
        final; its all-reduce overlaps the rest of the backward pass."""
        self._launch(lo, hi)

    def finish(self):
        """reduce everything not launched yet (PT: projector + task tokens; IFT: also the embeddings) and join (stream-side
        wait on the GPU)."""
        pos = 0
        for lo, hi in sorted(self.done) + [(self.g.numel(), self.g.numel())]:
            if lo > pos:
                self._launch(pos, lo)
            pos = max(pos, hi)
        for w in self.pending:
            w.wait()
        self.pending, self.done = [], []
