"""RCCL smoke on one GPU (world_size 1, backend nccl = RCCL): the exact torch.distributed calls the DP path issues
(async all_reduce on slices of the flat fp32 grad buffer, all_gather_into_tensor of bf16 targets) run and are no-ops
numerically.  Multi-rank semantics are covered on CPU by tests/test_parallel_gloo.py; the 8-GPU run is the driver's."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VP_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from visper_lm_amd import parallel
g = torch.arange(1000, device="cuda", dtype=torch.float32)
w1 = dist.all_reduce(g[0:640], async_op=True); w2 = dist.all_reduce(g[640:1000], async_op=True)
w1.wait(); w2.wait(); torch.cuda.synchronize()
assert torch.equal(g.cpu(), torch.arange(1000, dtype=torch.float32))
t = torch.randn(8, 1024, device="cuda").to(torch.bfloat16)
out = torch.empty(8, 1024, device="cuda", dtype=torch.bfloat16)
dist.all_gather_into_tensor(out, t); torch.cuda.synchronize()
assert torch.equal(out.cpu(), t.cpu())
x = torch.tensor([3.5], device="cuda", dtype=torch.float64); dist.all_reduce(x, op=dist.ReduceOp.MAX); dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
'''


NATIVE = r'''
import os, sys, torch
sys.path.insert(0, os.environ["VP_ROOT"])
torch.cuda.set_device(0)
from visper_lm_amd.parallel import NativeComm, GradReducer, all_gather_rows
c = NativeComm(rank=0, world=1)                       # vp_comm_unique_id + vp_comm_init: RCCL resolved inside libvisper_hip.so
g = torch.arange(4096, device="cuda", dtype=torch.float32)
g.mul_(2.0)                                           # queued on the compute stream BEFORE the reduce: the side stream must wait for it
c.allreduce_async(g[0:1024]); c.allreduce_async(g[1024:4096])
c.wait()                                              # the compute stream waits for both buckets (device-side)
g.add_(1.0)
torch.cuda.synchronize()
assert torch.equal(g.cpu(), torch.arange(4096, dtype=torch.float32) * 2 + 1)
b = torch.randn(8, 1000, device="cuda").to(torch.bfloat16)
for _ in range(70):                                   # more buckets than event slots: the ring of fences recycles
    c.allreduce_async(b)
c.wait(); torch.cuda.synchronize()
t = torch.randn(8, 1024, device="cuda").to(torch.bfloat16)
out = c.allgather(t); torch.cuda.synchronize()
assert out.shape == (8, 1024) and torch.equal(out.cpu(), t.cpu())
assert all_gather_rows(t, c) is t                     # world 1: no copy
r = GradReducer(g, split=100, comm=c); r.start_early(); r.finish()
c.close(); c.close()
print("NATIVE_COMM_OK")
'''


def test_native_comm_world1():
    """vp_comm_* (include/visper_hip.h) on one GPU: communicator creation through the C ABI, side-stream all-reduce fenced against the
    compute stream in both directions, all-gather, more in-flight buckets than fence slots, destroy.  Multi-rank numerics of the same
    call sequence are covered over gloo on the CPU; the 8-GPU run is the driver's."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", NATIVE], env=env, capture_output=True, text=True, timeout=240)
    assert "NATIVE_COMM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rccl_calls_world1():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", VP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=240)
    assert "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_under_torchrun_world1():
    """bench.py launched exactly like the driver does for N>1 (torch.distributed.run), with one rank and a shallow debug model."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29612", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--layers", "2",
           "--no-cpu-baseline", "--force-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert '"n_gpus": 1' in r.stdout and '"metric"' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_bench_two_ranks_on_one_gpu_control_flow():
    """The N>1 control flow of bench.py (per-rank data, barriers, MAX-over-ranks timing, one JSON line from rank 0, whole-job value)
    with two ranks sharing the test GPU over gloo (VP_TEST_SHARED_GPU) and a shallow debug model."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, VP_TEST_SHARED_GPU="1"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-1500:] + r.stderr[-2500:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 16 and res["scaling"] == "weak" and res["steps"] == 2
    assert abs(res["value"] - 16 / (res["ms_per_step"] / 1e3)) < 0.05 * res["value"]


def test_bench_starts_its_own_ranks():
    """plain `python bench.py --gpus 2` — NO torchrun around it — must start two ranks itself (bench.launch_ranks, like the
    reference's `deepspeed` line in scripts/train/pretrain.sh:15), and report n_gpus from the communicator.  Two ranks share the test GPU over
    gloo (VP_TEST_SHARED_GPU); on a real node the same path is one process per GPU over RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["VP_TEST_SHARED_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2", "--batch", "2",
           "--no-cpu-baseline", "--no-probes"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-2500:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["multi_gpu"]["torch_world_size"] == 2
    assert len(res["multi_gpu"]["per_rank_ms_per_step"]) == 2
    assert "exposed_comm_ms_per_step" in res["multi_gpu"][res["multi_gpu"]["headline_transport"]]


def test_bench_refuses_more_gpus_than_the_node_has():
    """`--gpus N` on a node with fewer GPUs fails loudly instead of running fewer ranks."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VP_TEST_SHARED_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout), r.stdout[-500:] + r.stderr[-500:]
    assert '"metric"' not in r.stdout


NATIVE2 = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VP_ROOT"])
rank = int(os.environ["RANK"])
multi = os.environ.get("VP_TEST_MULTI_DEVICE") == "1"      # >= 2 devices on the box: one device per rank, success REQUIRED
torch.cuda.set_device(rank if multi else 0)               # one device: BOTH ranks on it — RCCL must refuse (or serve) this without hanging
dist.init_process_group("nccl" if multi else "gloo", rank=rank, world_size=2,
                        **({"device_id": torch.device("cuda", rank)} if multi else {}))   # gloo only carries rank 0's unique id to rank 1
from visper_lm_amd.parallel import NativeComm, GradReducer, all_gather_rows
try:
    c = NativeComm(rank=rank, world=2)                 # vp_comm_unique_id -> broadcast -> vp_comm_init (collective)
    g = torch.full((1024,), float(rank + 1), device="cuda")
    c.allreduce_async(g); c.wait(); torch.cuda.synchronize()
    assert torch.equal(g.cpu(), torch.full((1024,), 3.0)), g[:4]
    t = torch.full((2, 8), float(rank), device="cuda").to(torch.bfloat16)
    out = c.allgather(t); torch.cuda.synchronize()
    assert out.shape == (4, 8) and float(out[0, 0]) == 0.0 and float(out[2, 0]) == 1.0
    if multi:
        # the native transport against the torch.distributed (nccl = RCCL) leg, bit for bit, on gradient-shaped data through the engine's
        # own GradReducer (bf16 buckets, early + late pieces) and the target all-gather
        gen = torch.Generator(device="cuda").manual_seed(100 + rank)
        base = torch.randn(3_000_000, device="cuda", generator=gen)
        res = {}
        for name, comm in (("torch", None), ("native", c)):
            buf = base.clone()
            r = GradReducer(buf, split=1_000_064, comm=comm, reduce_dtype=torch.bfloat16)
            r.start_early(); r.finish(); torch.cuda.synchronize()
            res[name] = buf
        assert torch.equal(res["torch"], res["native"]), "all-reduced gradients differ between the torch and the native transport"
        other = [torch.empty_like(res["native"]) for _ in range(2)]
        dist.all_gather(other, res["native"])
        assert torch.equal(other[0], other[1]), "all-reduced gradients differ across ranks"
        tg = torch.randn(8, 4096, device="cuda", generator=gen).to(torch.bfloat16)
        a, b = all_gather_rows(tg, None), all_gather_rows(tg, c)
        torch.cuda.synchronize()
        assert a.shape == (16, 4096) and torch.equal(a, b) and torch.equal(b[rank * 8:(rank + 1) * 8], tg)
        print(f"NATIVE2_RANK{rank}_BITWISE_OK")
    c.close()
    print(f"NATIVE2_RANK{rank}_RAN")
except RuntimeError as e:                              # the C ABI's error path: an int code turned into RuntimeError by _lib.call
    print(f"NATIVE2_RANK{rank}_REFUSED {str(e)[:200]}")
dist.destroy_process_group()
'''


def _multi_device():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def test_native_comm_two_processes_no_hang():
    """vp_comm_init with world 2 across two PROCESSES.  On a box with >= 2 devices: one device per rank over RCCL, success
    REQUIRED, and the native transport's all-reduced gradients / gathered targets must equal the torch.distributed leg's bit for bit and be
    identical on both ranks.  On the one-GPU test box both ranks sit on device 0: RCCL either refuses the duplicate device (-> the ABI's error
    code -> RuntimeError, the documented error path) or serves it; what must not happen is a hang or a crash."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    multi = _multi_device()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, VP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", RANK=str(r),
                   WORLD_SIZE="2", NCCL_DEBUG="WARN", VP_TEST_MULTI_DEVICE="1" if multi else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", NATIVE2], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("vp_comm_init with two processes did not return within 300 s (hang)")
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        print(f"rank {r}: rc={rc} {o.strip()[-300:]}")
        if multi:
            assert f"NATIVE2_RANK{r}_RAN" in o and f"NATIVE2_RANK{r}_BITWISE_OK" in o, (rc, o[-1500:], e[-1500:])
        else:
            assert f"NATIVE2_RANK{r}_RAN" in o or f"NATIVE2_RANK{r}_REFUSED" in o, (rc, o[-1500:], e[-1500:])


def test_bench_two_ranks_native_leg_is_the_headline_on_a_multi_gpu_box():
    """`python bench.py --gpus 2 --layers 4` the way a user types it.  With >= 2 devices: one device per rank over RCCL, BOTH transports must
    complete, the native (vp_comm_*) leg is the headline, every communicator saw two ranks, all-reduced gradients and gathered targets are
    bit-identical across ranks on both legs.  With one device: the same command on the shared-GPU hook (gloo), control flow only."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    multi = _multi_device()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "VP_TEST_SHARED_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if not multi:
        env["VP_TEST_SHARED_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--layers", "4", "--batch", "2",
           "--no-cpu-baseline", "--no-probes"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-2500:]
    res = json.loads(lines[0])
    mg = res["multi_gpu"]
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and mg["torch_world_size"] == 2
    if multi:
        assert mg["torch_backend"] == "nccl" and mg["headline_transport"] == "native", mg
        assert mg["n_ranks_seen"] == {"torch.distributed": 2, "vp_comm_info": 2}, mg["n_ranks_seen"]
        for leg in ("torch", "native"):
            assert "error" not in mg[leg], mg[leg]
            assert mg[leg]["grad_checksum_identical_across_ranks"] and mg[leg].get("target_checksum_identical_across_ranks", True), mg[leg]
        assert mg["torch"]["grad_checksum"] == mg["native"]["grad_checksum"], "the two transports reduce the same gradient to different bits"


def test_collective_on_the_comm_stream_beside_a_dynamic_gemm():
    """a vp_comm collective on the communicator's side stream BESIDE a many-round one-wave-per-SIMD GEMM with per-XCD dynamic
    tile claims (ops.set_dynamic(True): what world > 1 switches on; the counter blocks are caller-owned) — a self all-reduce at world 1, so it runs on any box.  The GEMM result
    must be bit-identical to the launch without the collective, round after round, and the reduced buffer intact."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import os, sys, torch
sys.path.insert(0, os.environ["VP_ROOT"])
torch.cuda.set_device(0)
from visper_lm_amd import ops
from visper_lm_amd.parallel import NativeComm
M, N, K = 16384, 4096, 1024                            # 1024 tiles = 4 rounds of the persistent grid
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
ref = ops.gemm(a, w); ref_r = ops.gemm(a, w, residual=res)
torch.cuda.synchronize()
c = NativeComm(rank=0, world=1)
buf = torch.randn(32 * 1024 * 1024, device="cuda", generator=g).to(torch.bfloat16)       # 64 MB bucket
keep = buf.clone()
prev = ops.set_dynamic(True)
try:
    for it in range(6):
        c.allreduce_async(buf)                         # side stream, fenced behind the compute stream's work so far
        o1 = ops.gemm(a, w); o2 = ops.gemm(a, w, residual=res)
        c.allreduce_async(buf)
        o3 = ops.gemm(a, w)
        c.wait()
        assert torch.equal(o1, ref) and torch.equal(o2, ref_r) and torch.equal(o3, ref), f"round {it}"
    torch.cuda.synchronize()
    assert torch.equal(buf, keep)                      # world 1: the sum over one rank
finally:
    ops.set_dynamic(prev)
c.close()
print("COMM_BESIDE_GEMM_OK")
"""
    env = dict(os.environ, VP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert "COMM_BESIDE_GEMM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_w4_gemms_beside_1000_rccl_collectives_world1():
    """the one-wave-per-SIMD GEMM (all three epilogue instantiations: lean, general with bias / activation / residual and an M
    tail, RMSNorm-fold with sums of squares) beside RCCL kernels on the communicator's side stream for >= 1000 GEMM launches: self all-reduces of a
    16 MB bucket at world 1, issued back to back so that a collective kernel is resident for the whole run (the next co-resident party after the
    side stream's small kernels, and on CUs the GEMM's register-file claim does not cover).  Every output must equal the bits of the solo launch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import os, sys, torch
sys.path.insert(0, os.environ["VP_ROOT"])
torch.cuda.set_device(0)
from visper_lm_amd import ops
from visper_lm_amd.parallel import NativeComm
M, N, K = 8192, 4096, 1024
g = torch.Generator(device="cuda").manual_seed(7)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
Mt = M - 100
def launch():
    return (ops.gemm(a, w), ops.gemm(a, w, bias=bias, residual=res, epi=ops.EPI_QUICK_GELU, force_generic=14),
            ops.gemm(a[:Mt], w, bias=bias, force_generic=14), *ops.gemm_sumsq(a, w, res))
solo = launch()
torch.cuda.synchronize()
c = NativeComm(rank=0, world=1)
buf = torch.randn(8 * 1024 * 1024, device="cuda", generator=g).to(torch.bfloat16)         # 16 MB bucket
keep = buf.clone()
n = 0
for it in range(260):                                  # 260 x 4 = 1040 GEMM launches
    c.allreduce_async(buf); c.allreduce_async(buf)     # two collectives in flight beside this round's four GEMMs
    got = launch()
    n += 4
    if it % 10 == 9:
        c.wait()
    for i, (g_, s_) in enumerate(zip(got, solo)):
        assert torch.equal(g_, s_), f"round {it}, output {i}: differs from the solo launch"
c.wait()
torch.cuda.synchronize()
assert torch.equal(buf, keep) and n >= 1000
c.close()
print("W4_BESIDE_RCCL_OK", n)
"""
    env = dict(os.environ, VP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
    assert "W4_BESIDE_RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
