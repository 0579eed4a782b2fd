"""Data-parallel exchange points of the PT step.  One process per GPU; exactly two kinds of collective per step:

  1. all_gather_rows: the frozen-teacher target features, once per task per step (the reference re-gathers the normalised
     targets in every layer call: ola_utils.py:96-106,118-119) -> rank-ordered [world*B, D].
  2. GradReducer: sum-all-reduce of the flat gradient buffer in buckets — heads + logit scales as soon as the heads' backward is
     done (their all-reduce runs on the communication stream underneath the whole decoder backward), projector + task tokens at
     the end; IFT stage: one bucket per decoder layer in backward order.  The 1/world mean is folded into the fused AdamW
     (grad_scale).  Replaces DeepSpeed ZeRO-2's bucketed reduce-scatter (scripts/zero2.json:16-22, overlap_comm:false).
     Buckets travel in bf16 by default, like the reference's (DeepSpeed reduces the bf16 gradients of a bf16 model and adds them to
     fp32 partitions): half the bytes per xGMI link; `reduce_dtype=torch.float32` keeps fp32 on the wire.

Two transports with the same semantics:
  * "torch"  — torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests); the default.
  * "native" — the C ABI's own communicator (csrc/comm.hip: RCCL + a side HIP stream + hipEvent fences inside libvisper_hip.so,
     include/visper_hip.h vp_comm_*), for callers without PyTorch collectives; torch.distributed is used only to hand rank 0's
     unique id to the other ranks.  Select with Engine.set_distributed(..., transport="native") or VP_COMM=native.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2}


class NativeComm:
    """vp_comm_* communicator (RCCL resolved at run time inside libvisper_hip.so).  Collective constructor."""

    def __init__(self, rank=None, world=None, unique_id: bytes = None):
        from . import _lib
        self._lib = _lib
        r, w = world_info()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        n = _lib.raw("vp_comm_unique_id_bytes")
        if unique_id is None:
            buf = (C.c_ubyte * n)()
            if self.rank == 0:
                _lib.call("vp_comm_unique_id", buf)
            if self.world > 1:
                box = [bytes(buf)]
                dist.broadcast_object_list(box, src=0)          # out-of-band id exchange (any backend; a file or MPI works as well)
                unique_id = box[0]
            else:
                unique_id = bytes(buf)
        self.h = C.c_void_p()
        idbuf = (C.c_ubyte * n).from_buffer_copy(unique_id)
        _lib.call("vp_comm_init", self.rank, self.world, idbuf, C.byref(self.h))

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def allreduce_async(self, t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DT
        self._lib.call("vp_comm_allreduce_async", self.h, C.c_void_p(t.data_ptr()), t.numel(), _DT[t.dtype], self._stream())

    def wait(self):
        self._lib.call("vp_comm_wait", self.h, self._stream())

    def allgather(self, flat: torch.Tensor) -> torch.Tensor:
        flat = flat.contiguous()
        out = torch.empty(self.world * flat.shape[0], *flat.shape[1:], device=flat.device, dtype=flat.dtype)
        self._lib.call("vp_comm_allgather", self.h, C.c_void_p(flat.data_ptr()), C.c_void_p(out.data_ptr()), flat.numel(), _DT[flat.dtype],
                       self._stream())
        return out

    def close(self):
        if self.h:
            self._lib.call("vp_comm_destroy", self.h)
            self.h = C.c_void_p()


def all_gather_rows(flat: torch.Tensor, comm: NativeComm = None, dry_world: int = 0) -> torch.Tensor:
    """[B, D] on every rank -> [world*B, D], rank r's rows at r*B..(r+1)*B (dist_collect order, ola_utils.py:104-106).
    dry_world > 0 (bench.py's "no communication" leg): no collective, the local rows repeated dry_world times (same shapes downstream)."""
    if dry_world > 0:
        return flat.repeat(dry_world, 1) if dry_world > 1 else flat
    if comm is not None:
        return comm.allgather(flat) if comm.world > 1 else flat
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
    def __init__(self, flat_grad: torch.Tensor, split: int, comm: NativeComm = None, reduce_dtype=torch.bfloat16):
        self.g, self.split, self.comm = flat_grad, split, comm
        self.pending = []                               # torch.distributed work handles
        self.done = []                                  # [lo, hi) ranges already launched this step
        self.reduce_dtype = reduce_dtype
        self._stage = None                              # bf16 wire copy of the gradient buffer (allocated on first use)
        self.dry = False                                # bench.py's "no communication" leg: buckets are tracked but nothing goes on the wire

    def _world(self):
        return self.comm.world if self.comm is not None else world_info()[1]

    def _wire(self, lo, hi):
        """The tensor that goes on the wire for bucket [lo, hi): the fp32 gradients themselves, or their bf16 cast."""
        if self.reduce_dtype == torch.float32:
            return self.g[lo:hi]
        if self._stage is None:
            self._stage = torch.empty(self.g.numel(), device=self.g.device, dtype=self.reduce_dtype)
        st = self._stage[lo:hi]
        if self.g.is_cuda:
            from . import ops
            ops.cast_to_bf16(self.g[lo:hi], out=st)
        else:
            st.copy_(self.g[lo:hi])
        return st

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        self.done.append((lo, hi))
        if self._world() <= 1 or self.dry:
            return
        w = self._wire(lo, hi)
        if self.comm is not None:
            self.comm.allreduce_async(w)
        elif dist.get_backend() == "gloo" and w.is_cuda:          # CPU test backend with device tensors: stage through the host
            h = w.float().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            w.copy_(h.to(w.dtype))
        else:
            self.pending.append(dist.all_reduce(w, op=dist.ReduceOp.SUM, async_op=True))

    def start_early(self):
        """heads + logit-scale gradients are final: reduce them while the decoder backward runs."""
        self._launch(0, self.split)

    def reduce_range(self, lo, hi):
        """IFT stage: one bucket (lm_head + final norm, then one decoder layer at a time, in backward order) whose gradients are
        final; its all-reduce overlaps the rest of the backward pass."""
        self._launch(lo, hi)

    def finish(self):
        """reduce everything not launched yet (PT: projector + task tokens; IFT: also the embeddings), join (stream-side wait on the
        GPU; the host does not block) and, for bf16 buckets, add the reduced values back into the fp32 gradient buffer."""
        pos = 0
        for lo, hi in sorted(self.done) + [(self.g.numel(), self.g.numel())]:
            if lo > pos:
                self._launch(pos, lo)
            pos = max(pos, hi)
        for w in self.pending:
            w.wait()
        if self.comm is not None and self._world() > 1 and not self.dry:
            self.comm.wait()
        if self.reduce_dtype != torch.float32 and self._world() > 1 and not self.dry:
            for lo, hi in self.done:
                if self.g.is_cuda:
                    from . import ops
                    ops.cast_to_f32(self._stage[lo:hi], out=self.g[lo:hi])
                else:
                    self.g[lo:hi].copy_(self._stage[lo:hi])
        self.pending, self.done = [], []
