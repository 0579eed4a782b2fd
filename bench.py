"""bench.py — train-step images/sec of the VisPer-LM PT step (NTP + distillation) on MI355X.

Workload (BASELINE.json configs[1]): CLIP-ViT-L/14-336 + Llama-3-8B, bf16, batch 8 per GPU,
post-splice seq_len 2048 (text T=1449, one 336x336 image at column 38, 3x8 task tokens), one distillation head per
task (depth@18, seg@18, gen@20), random-init weights and synthetic data/targets resident in HBM.
A step = ViT fwd + projector fwd/bwd + splice + 32-layer decoder fwd + dgrad bwd + lm_head/CE fwd/bwd + 3 heads
fwd/bwd + embedding losses + (N>1: RCCL grad all-reduce, contrastive target all-gather) + fused AdamW.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the bf16 MFMA
GEMM, timed live with HIP events on its launch stream) and `cpu_baseline` (the CPU oracle timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

STEP_TF_PER_IMAGE = 65.84       # SURVEY §8d / BASELINE.md §2: algorithmic PT-step TFLOP per image (config 2)
PEAK_BF16_TF = 2500.0           # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def make_batch(cfg, B, T, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    ids = torch.randint(0, 1000, (B, T), generator=g)
    ids[:, cfg.num_sys_tokens] = -200
    labels = ids.clone()
    labels[:, :cfg.num_sys_tokens + 7] = -100
    gd = torch.Generator(device=device).manual_seed(1234 + rank)
    rn = lambda *s: torch.randn(*s, device=device, dtype=torch.bfloat16, generator=gd)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids, dtype=torch.bool),
                 images=rn(B, 3, cfg.cnx_image if cfg.is_convnext else cfg.vit_image, cfg.cnx_image if cfg.is_convnext else cfg.vit_image))
    order = cfg.token_order
    if "gen" in order:
        batch["gen_target"] = rn(B, 1, cfg.image_gen["output_dim"]); batch["gen_mask"] = torch.ones(B, device=device)
    if "depth" in order:
        batch["depth_target"] = rn(B, 576, cfg.image_depth["output_dim"]); batch["depth_mask"] = torch.ones(B, device=device)
    if "seg" in order:
        batch["seg_target"] = rn(B, cfg.image_seg["output_dim"], 24, 24); batch["seg_mask"] = torch.ones(B, device=device)
    return batch


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, dev):
    """The CPU oracle (a port of the reference's PyTorch path; oracle/visper_oracle.py) timed on this box's host cores on a BOUNDED
    sample of the same workload: REAL, WHOLE fwd + bwd steps of the PT path at configs[0] shapes with the headline's architecture at FULL
    depth (B = 2 images, text 128 -> S = 727: full CLIP-ViT-L tower, projector fwd/bwd, splice, all 32 full-width Llama-3-8B decoder layers
    fwd + dgrad, lm_head + CE at V = 128256, the three distillation heads d18 / s18 / g20 fwd/bwd + losses), bf16 like the reference's CPU
    path.  One warm-up step, then the MEDIAN of two timed steps; nothing is extrapolated.  Fixed thread count (16, or every core of a
    smaller host): on the pool's 256-thread EPYC 9575F hosts the GEMMs of a 1454-token batch run fastest there (rounds 3-5 probed 16 / 64 /
    256 threads in every run: 0.28 / 0.50 / 8.7 s per decoder layer forward) and a per-run choice was one source of the 2x run-to-run
    spread of the earlier, layer-extrapolated figure."""
    from oracle import visper_oracle as O
    from visper_lm_amd.config import llama3_8b
    from visper_lm_amd.engine import is_trainable
    from visper_lm_amd.params import param_shapes, init_value
    ncpu = os.cpu_count() or 1
    dt = torch.bfloat16
    B, T = 2, 128
    th = min(ncpu, int(os.environ.get("VP_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(th)
    c = llama3_8b(num_hidden_layers=cfg.num_hidden_layers)
    c.image_seg, c.image_depth, c.image_gen = dict(cfg.image_seg), dict(cfg.image_depth), dict(cfg.image_gen)
    gen = torch.Generator(device=dev).manual_seed(1)
    W = {}
    for k, shp in param_shapes(c, vit_nested=True).items():
        if k.startswith("da_v2_head."):
            continue
        w = init_value(k, shp, gen, dev, dt if len(shp) else torch.float32).cpu()      # generated on the GPU, copied once
        W[k] = w.requires_grad_(True) if is_trainable(k) else w
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 1000, (B, T), generator=g)
    ids[:, c.num_sys_tokens] = -200
    lab = ids.clone()
    lab[:, :c.num_sys_tokens + 7] = -100
    rn = lambda *s_: torch.randn(*s_, generator=g).to(dt)
    batch = dict(input_ids=ids, labels=lab, attention_mask=torch.ones_like(ids, dtype=torch.bool), images=rn(B, 3, 336, 336),
                 gen_target=rn(B, 1, 1024), gen_mask=torch.ones(B), depth_target=rn(B, 576, 1024), depth_mask=torch.ones(B),
                 seg_target=rn(B, 1536, 24, 24), seg_mask=torch.ones(B))
    ocfg = O.make_config(**{k: v for k, v in c.to_dict().items() if k in vars(O.make_config())})

    def timed_step():
        t0 = time.time()
        out = O.forward(W, batch, ocfg, need_logits=False)
        out["loss"].backward()
        el = time.time() - t0
        for w in W.values():
            w.grad = None
        return el, float(out["loss"])

    t_warm, _ = timed_step()                                   # warm-up (allocator, oneDNN primitive caches)
    runs = [timed_step() for _ in range(2)]
    ts = sorted(t for t, _ in runs)
    step = 0.5 * (ts[0] + ts[1])
    return {"value": round(B / step, 5), "unit": "images/s", "cores": th, "host_cpus": ncpu, "cpu_model": _cpu_model(), "dtype": "bf16",
            "kind": "port", "extrapolated": False, "step_s": [round(t, 2) for t, _ in runs], "warmup_step_s": round(t_warm, 2),
            "sample": (f"oracle bf16, {th} threads ({_cpu_model()}, {ncpu} logical CPUs): whole fwd+bwd steps of the PT path at configs[0] shapes "
                       f"(B=2, text 128 -> S=727, ViT-L full, ALL {c.num_hidden_layers} decoder layers, 3 heads, V=128256); 1 warm-up step "
                       f"({t_warm:.1f}s), median of 2 timed steps {runs[0][0]:.1f}s / {runs[1][0]:.1f}s (loss {runs[1][1]:.4f}); nothing extrapolated")}


def run_extras(args, headline_ms, step_tf_hint=None):
    """Secondary legs of the default line, each a FRESH process of this script started after the headline's timed region (this process has
    freed the GPU by then): W = 2 warm-up + K = --extras-steps timed steps between the same barrier + synchronize fences, every other probe
    off.  engine_direct = Engine.train_step / optimizer_step without the model class (rounds 1-5's headline); reference_outputs = the model
    class with its contract default config.reference_outputs=True (all L + 1 layer states returned, fp32 logits of all B x S rows handed out
    lazily: ola_llama.py:113-122) driven as a trainer drives it (reads .loss); ..._logits_read = the same with out.logits read every step;
    configs[4] / configs[3] = the other two single-GPU configurations of BASELINE.json through the model class."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.extras_steps), "--warmup", "2", "--no-probes",
            "--no-cpu-baseline", "--no-extras", "--lr", str(args.lr)] + (["--no-depth-decoder"] if args.no_depth_decoder else [])
    legs = [("engine_direct", ["--api", "engine"]), ("reference_outputs", ["--api", "model", "--reference-outputs"]),
            ("reference_outputs_logits_read", ["--api", "model", "--reference-outputs", "--read-logits"]),
            ("configs[4]_phi3", ["--api", "model", "--workload", "phi3"]), ("configs[3]_convnext", ["--api", "model", "--workload", "convnext"])]
    only = os.environ.get("VP_BENCH_EXTRAS")
    out = {"what": run_extras.__doc__.split("\n")[0].strip() + " ... (bench.py run_extras)", "steps": args.extras_steps, "warmup": 2}
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    for tag, extra in legs:
        if only and tag not in only.split(","):
            continue
        t0 = time.time()
        try:
            pr = subprocess.run(base + extra, capture_output=True, text=True, cwd=ROOT, env=env,
                                timeout=float(os.environ.get("VP_BENCH_EXTRA_TIMEOUT", "900")))
            lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if not lines:
                raise RuntimeError(f"no result line (exit code {pr.returncode}): {(pr.stderr or pr.stdout)[-300:]}")
            r = json.loads(lines[-1])
            rf, cf = r["roofline"], r["config"]
            out[tag] = {"ms_per_step": r["ms_per_step"], "images_per_s": r["value"], "step_frac_of_peak_executed_flops": rf.get("step_frac_of_peak"),
                        "executed_tflop_per_step": rf.get("executed_tflop_per_step"), "gemm_frac": rf.get("frac"),
                        "gemm_family_frac": rf.get("family", {}).get("frac"), "loss": cf.get("loss"), "api": cf.get("api"), "seq_len": cf.get("seq_len"),
                        "per_gpu_batch": cf.get("per_gpu_batch"), "peak_mem_gb": cf.get("peak_mem_gb"), "lm_head_rows": cf.get("lm_head_rows"),
                        "outputs": cf.get("outputs", "")[:60], "workload": cf.get("workload"), "process_wall_s": round(time.time() - t0, 1)}
            if tag in ("engine_direct", "reference_outputs", "reference_outputs_logits_read"):
                out[tag]["delta_ms_vs_headline"] = round(r["ms_per_step"] - headline_ms, 2)
        except Exception as e:                              # noqa: BLE001  (a secondary leg must never cost the headline line)
            out[tag] = {"error": f"{type(e).__name__}: {e}"[:400]}
    return out


def clock_probe(dev, n=120):
    """Sustained shader clock of the dominant kernel: `n` back-to-back launches of its longest decoder shape (the package sits at its power
    cap, like inside the step), the last one with in-kernel stamps (s_memtime wall clock + shader-cycle counter around the first output tile's
    K loop of every block, vp_debug_gemm_flags).  Returns MHz, K-loop cycles per 64-wide K-tile (2048 = pure MFMA issue for a 256x256 tile on
    4 SIMDs), and the per-XCD clocks (each XCD is its own DVFS domain; the slowest one sets the kernel time)."""
    import ctypes as C
    import numpy as np
    from visper_lm_amd import ops, _lib
    M, N, K = 16384, 4096, 14336
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    # (the in-kernel stamps exist in the -DVP_DEBUG build of the library only: the probe's launches go to libvisper_hip_debug.so, same kernels)
    with _lib.debug_library():
        for _ in range(n - 1):
            ops.gemm(a, w, out=o)
        _lib.call("vp_debug_gemm_flags", 0x10000)
        try:
            ops.gemm(a, w, out=o)
            torch.cuda.synchronize()
        finally:
            _lib.call("vp_debug_gemm_flags", 0)
        buf = (C.c_long * 2048)()
        _lib.call("vp_debug_stamps", buf)
    st = np.array(buf[:], dtype=np.int64).reshape(256, 8)
    us = (st[:, 2] - st[:, 1]) / 100.0                          # wall clock ticks are 10 ns
    cyc = (st[:, 7] - st[:, 6]).astype(np.float64)
    mhz = cyc / us
    return {"shape_MNK": [M, N, K], "shader_clock_mhz": round(float(mhz.mean()), 0), "nominal_mhz": 2400,
            "per_xcd_mhz": [int(mhz[x::8].mean()) for x in range(8)],
            "k_loop_cycles_per_k_tile": round(float(cyc.mean()) / (K // 64), 1), "mfma_issue_cycles_per_k_tile": 2048,
            "mfma_issue_util_in_k_loop": round(2048.0 * (K // 64) / float(cyc.mean()), 4)}


def k11_probe(cfg, B, world, dev, n=25):
    """The distillation-loss reduction (vp_emb_loss_fwd / _bwd; base_ola_vlm.py:289-320, ola_utils.py:108-125) timed alone with HIP events
    on its launch stream at this workload's shapes, against the 8 TB/s HBM roofline.  Algorithmic bytes: forward reads pred
    + gathered targets once = 2*D*(B + B*world); backward re-reads them and writes dpred = 2*D*(2B + B*world)."""
    from visper_lm_amd import ops
    out = {}
    use_graph = [not (torch.distributed.is_available() and torch.distributed.is_initialized())]

    def t_loop(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def t(fn, reps=4):
        # n back-to-back launches captured ONCE into a HIP graph and replayed: device time per launch including the launch gap, without the
        # host's allocation + ctypes time per call (10-30 us in python: more than the kernel takes, so a python loop would time the host).
        # With a process group alive (N > 1) nothing is captured — RCCL's watchdog thread may touch the device during a capture — and the
        # python loop's (host-bound, pessimistic) number is reported instead; any capture failure falls back the same way.
        if not use_graph[0]:
            return t_loop(fn)
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    for _ in range(n):
                        fn()
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(reps):
                    g.replay()
                e1.record(s); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (n * reps) * 1e3
        except Exception:                               # noqa: BLE001  (a probe must never take the bench line down)
            use_graph[0] = False
            torch.cuda.synchronize()
            return t_loop(fn)

    for task, D in (("gen", cfg.image_gen["output_dim"]), ("depth", 576 * cfg.image_depth["output_dim"]), ("seg", 576 * cfg.image_seg["output_dim"])):
        if task not in cfg.token_order:
            continue
        Bw = B * world
        pred = torch.randn(B, D, device=dev, dtype=torch.bfloat16)
        tgt = torch.randn(Bw, D, device=dev, dtype=torch.bfloat16)
        mask = torch.ones(B, device=dev)
        scale = torch.full((1,), 2.0, device=dev)
        _, coef = ops.emb_loss_fwd(pred, tgt, mask, scale, 0.3)

        uf = t(lambda: ops.emb_loss_fwd(pred, tgt, mask, scale, 0.3))
        ub = t(lambda: ops.emb_loss_bwd(pred, tgt, coef, 0.5))
        bf, bb = 2.0 * D * (B + Bw), 2.0 * D * (2 * B + Bw)
        out[task] = {"D": D, "fwd_us": round(uf, 1), "bwd_us": round(ub, 1), "fwd_GBps": round(bf / uf / 1e3, 1), "bwd_GBps": round(bb / ub / 1e3, 1),
                     "fwd_frac_of_8TBps": round(bf / uf / 1e3 / 8000.0, 3), "bwd_frac_of_8TBps": round(bb / ub / 1e3 / 8000.0, 3),
                     "fwd_launches": 1, "bwd_launches": 1}
    # every head of the step in ONE launch each way (vp_emb_loss_{fwd,bwd}_multi: what the engine issues): bytes of all tasks / one launch
    tasks = [(t, out[t]["D"]) for t in ("depth", "seg", "gen") if t in out]
    if len(tasks) > 1:
        Bw = B * world
        preds = [torch.randn(B, D, device=dev, dtype=torch.bfloat16) for _, D in tasks]
        tgts = [torch.randn(Bw, D, device=dev, dtype=torch.bfloat16) for _, D in tasks]
        masks = [torch.ones(B, device=dev) for _ in tasks]
        scales = [torch.full((1,), 2.0, device=dev) for _ in tasks]
        res = ops.emb_loss_fwd_multi(preds, tgts, masks, scales, [0.3] * len(tasks))
        coefs = [c for _, c in res]

        uf = t(lambda: ops.emb_loss_fwd_multi(preds, tgts, masks, scales, [0.3] * len(tasks)))
        ub = t(lambda: ops.emb_loss_bwd_multi(preds, tgts, coefs, [0.5] * len(tasks)))
        bf = sum(2.0 * D * (B + Bw) for _, D in tasks); bb = sum(2.0 * D * (2 * B + Bw) for _, D in tasks)
        out["timed_by"] = "hip_graph_replay" if use_graph[0] else "python_loop (host-bound)"
        out["all_heads_one_launch"] = {"tasks": [t for t, _ in tasks], "fwd_us": round(uf, 1), "bwd_us": round(ub, 1),
                                       "fwd_GBps": round(bf / uf / 1e3, 1), "bwd_GBps": round(bb / ub / 1e3, 1),
                                       "fwd_frac_of_8TBps": round(bf / uf / 1e3 / 8000.0, 3), "bwd_frac_of_8TBps": round(bb / ub / 1e3 / 8000.0, 3),
                                       "fwd_launches": 1, "bwd_launches": 1}
    return out


def _checksum(t):
    """(sum, sum of |x|, xor of the raw bits) of a device tensor as python numbers: equal on every rank iff the tensors are bit-identical
    (up to xor / sum collisions).  Bench-side diagnosis only."""
    f = t.detach().reshape(-1)
    bits = f.view(torch.int32 if f.element_size() == 4 else torch.int16).to(torch.int64)
    x = 0
    for chunk in bits.split(1 << 24):
        v = chunk
        while v.numel() > 1:                           # xor tree (torch has no xor reduction)
            if v.numel() % 2:
                v = torch.cat([v, v.new_zeros(1)])
            v = v[0::2] ^ v[1::2]
        x ^= int(v[0])
    return [float(f.double().sum()), float(f.double().abs().sum()), x]


def multi_gpu_report(args, eng, dist, dev, rank, world, legs, dt_own, fresh, step, timed_leg, max_over_ranks, fence, shared):
    """Self-diagnosis of an N > 1 run (the builder cannot rehearse it): did RCCL see N ranks, are the all-reduced gradients and the gathered
    targets bit-identical on every rank, how long do the two exchange points take alone, what does a step cost with the communication
    switched off (exposed communication = step - that), and the same K timed steps over the OTHER transport."""
    import threading
    from visper_lm_amd.parallel import all_gather_rows
    B = args.batch
    per_rank = [None] * world
    dist.all_gather_object(per_rank, round(dt_own / args.steps * 1e3, 2))
    diag = {"primary_transport": legs[0], "per_rank_ms_per_step": per_rank, "torch_world_size": dist.get_world_size(),
            "torch_backend": dist.get_backend()}
    assert dist.get_world_size() == world == args.gpus or shared, (dist.get_world_size(), world, args.gpus)

    def timed(fn, n=5):
        fn(); fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / n, 3)

    def exchange_points(tag):
        """both collectives alone (blocking, HIP events) + cross-rank checksums of their results, for the transport currently set"""
        red = eng._reducer()

        def reduce_all():
            red.start_early(); red.finish()
        seg_t = next((fresh[0][k] for k in ("seg_target", "depth_target", "gen_target") if fresh[0].get(k) is not None), None)
        d = {"grad_allreduce_ms_alone": timed(reduce_all),
             "grad_bytes_on_wire": int(eng.ps.grad.numel() * (2 if red.reduce_dtype == torch.bfloat16 else 4)),
             "grad_reduce_dtype": str(red.reduce_dtype).replace("torch.", "")}
        if eng.comm is not None:
            import ctypes as C
            r_, w_ = C.c_int(-1), C.c_int(-1)
            eng.comm._lib.call("vp_comm_info", eng.comm.h, C.byref(r_), C.byref(w_))
            d["vp_comm_info"] = {"rank": r_.value, "world": w_.value}
            assert w_.value == world, (w_.value, world)
        # a deterministic per-rank gradient -> all-reduce -> every rank must hold the same bits
        g = eng.ps.grad
        g.copy_(torch.sin(torch.arange(g.numel(), device=dev, dtype=torch.float32) * 1e-3 + rank))
        red.start_early(); red.finish(); torch.cuda.synchronize()
        cs = [None] * world
        dist.all_gather_object(cs, _checksum(g))
        d["grad_checksum_identical_across_ranks"] = all(c == cs[0] for c in cs)
        d["grad_checksum"] = cs[0]
        if seg_t is not None:
            flat = seg_t.reshape(B, -1).contiguous()
            d["target_allgather_ms_alone"] = timed(lambda: all_gather_rows(flat, eng.comm))
            allt = all_gather_rows(flat, eng.comm); torch.cuda.synchronize()
            ct = [None] * world
            dist.all_gather_object(ct, _checksum(allt))
            d["target_checksum_identical_across_ranks"] = all(c == ct[0] for c in ct)
            own = allt[rank * B:(rank + 1) * B]
            d["own_rows_at_rank_offset"] = bool(torch.equal(own, flat))
            d["gathered_rows"] = int(allt.shape[0])
        assert d["grad_checksum_identical_across_ranks"], f"{tag}: all-reduced gradients differ across ranks: {cs}"
        assert d.get("target_checksum_identical_across_ranks", True), f"{tag}: gathered targets differ across ranks"
        return d

    diag[legs[0]] = dict(ms_per_step=round(max_over_ranks(dt_own) / args.steps * 1e3, 2), **exchange_points(legs[0]))
    # the same steps with nothing on the wire: exposed communication = step - this
    eng.comm_dry = True
    eng._red = None
    el, _, _ = timed_leg(1, args.steps)
    eng.comm_dry = False
    eng._red = None
    dry_ms = max_over_ranks(el) / args.steps * 1e3
    diag["ms_per_step_without_communication"] = round(dry_ms, 2)
    diag[legs[0]]["exposed_comm_ms_per_step"] = round(diag[legs[0]]["ms_per_step"] - dry_ms, 2)
    diag["_ctx"] = (exchange_points, dry_ms)
    return diag


def alt_transport_legs(args, eng, dist, rank, world, legs, diag, timed_leg, max_over_ranks, emit_partial):
    """The same K timed steps over the OTHER transport(s), behind a watchdog: the native communicator has never met N > 1 ranks before the
    driver's run, and a hang inside its init must not take the primary result down — after VP_BENCH_ALT_TIMEOUT seconds (default 180) rank 0
    prints the line it already has (the leg marked as timed out) and every rank exits."""
    import threading
    exchange_points, dry_ms = diag.pop("_ctx")
    for alt in legs[1:]:
        bail = threading.Event()

        def watchdog():
            if not bail.wait(float(os.environ.get("VP_BENCH_ALT_TIMEOUT", "180"))):
                diag[alt] = {"error": "timeout: the leg did not finish; primary result unaffected"}
                if rank == 0:
                    emit_partial()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            eng.set_distributed(rank, world, transport=alt)
            el, _, _ = timed_leg(max(1, args.warmup), args.steps)
            diag[alt] = dict(ms_per_step=round(max_over_ranks(el) / args.steps * 1e3, 2), **exchange_points(alt))
            diag[alt]["exposed_comm_ms_per_step"] = round(diag[alt]["ms_per_step"] - dry_ms, 2)
        except Exception as e:                          # noqa: BLE001  (a failed alternate leg is a reported result, not a crash)
            diag[alt] = {"error": f"{type(e).__name__}: {e}"[:500]}
        finally:
            bail.set()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks OURSELVES, one process per GPU, the way the
    reference's own script does (scripts/train/pretrain.sh:15 `deepspeed ...` spawns the ranks; the caller never does).  The parent holds
    no GPU: it re-runs this file under torch.distributed.run on 127.0.0.1 with a free port, relays the ranks' stderr, and prints the ONE
    JSON line rank 0 produced.  Fails loudly when the node has fewer than N GPUs (VP_TEST_SHARED_GPU=1, the one-GPU test hook, excepted)."""
    import signal
    import subprocess
    n = args.gpus
    shared = os.environ.get("VP_TEST_SHARED_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not shared:
        raise SystemExit(f"bench.py --gpus {n}: this node exposes {have} GPU(s); refusing to run fewer ranks than asked for")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1",
               VP_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    limit = float(os.environ.get("VP_BENCH_LAUNCH_TIMEOUT", "2400"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True, cwd=ROOT)
    try:
        out, _ = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)                    # the process group WE started (start_new_session): nothing else
        out, _ = p.communicate()
        sys.stderr.write(f"bench.py: the {n}-rank job did not finish within {limit:.0f} s and was killed\n")
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
    if not lines:
        sys.stderr.write((out or "")[-4000:])
        raise SystemExit(f"bench.py --gpus {n}: the ranks produced no result line (exit code {p.returncode})")
    res = json.loads(lines[-1])
    if res.get("n_gpus") != n:
        raise SystemExit(f"bench.py --gpus {n}: the communicator reported {res.get('n_gpus')} ranks: {lines[-1][:400]}")
    if p.returncode not in (0, None):                      # a complete, checked line exists: a crash during teardown does not void it
        sys.stderr.write(f"bench.py: ranks exited with code {p.returncode} after producing the result line\n")
    print(lines[-1], flush=True)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--text-len", type=int, default=1449)          # -> post-splice S = 2048 with 3 tasks
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer decoder layers (result marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true", help="skip the stand-alone K11 / shader-clock probes (profiling runs: keeps their launches out of the kernel stats)")
    ap.add_argument("--no-depth-decoder", action="store_true",
                    help="skip the frozen DPT depth decoder (depth_preds: a logging-only output the reference computes under no_grad in "
                         "every training step, base_ola_vlm.py:462-470; ~6 ms/step here); on by default so the timed step does all the "
                         "reference's work")
    ap.add_argument("--workload", default="llama3_8b", choices=["llama3_8b", "convnext", "phi3", "ift", "pt6"],
                    help="llama3_8b = BASELINE configs[1] (the headline metric); convnext = configs[3]; phi3 = configs[4]; ift = SURVEY f-2; "
                         "pt6 = the reference's script-default recipe, 6 heads d18-20_s10-18_g12-20 (scripts/train/pretrain.sh:20) (secondary); "
                         "the script's per-device batch is --batch 32 (pretrain.sh:38)")
    ap.add_argument("--transports", default="both", choices=["both", "torch", "native"],
                    help="N > 1: which DP transports to time (the JSON line's value is the first one's; the other is reported under multi_gpu)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--reference-outputs", action="store_true",
                    help="materialise what the reference's forward returns every step (logits of all B x S rows + every layer state: the mirror's "
                         "config.reference_outputs=True default) instead of the lean training mode (loss only, lm_head on labelled rows)")
    ap.add_argument("--api", default="model", choices=["model", "engine"],
                    help="model (default) = the drop-in boundary north_star names: every step is `out = model(**batch); out.loss.backward(); "
                         "model.optimizer_step(lr)` through the reference-named class (OlaLlavaLlamaForCausalLM / OlaLlavaPhi3ForCausalLM / "
                         "LlavaLlamaForCausalLM: ola_vlm_train.py:1297-1327 -> ola_llama.py:190-244); engine = Engine.train_step + "
                         "Engine.optimizer_step called directly (what rounds 1-5 timed; reported beside the headline under `extras`)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs of the default line (engine-direct, reference outputs, configs[4] phi3, configs[3] convnext: each a "
                         "fresh process of this script after the timed region, a few steps between the same fences)")
    ap.add_argument("--extras-steps", type=int, default=3)
    ap.add_argument("--read-logits", action="store_true",
                    help="with --reference-outputs: also READ out.logits every step (the training forward hands them out lazily: a trainer that only "
                         "reads .loss never pays for the fp32 [B, S, V] tensor; this flag times the step of a caller that reads it every step and "
                         "therefore asks for it with output_logits=True: label-less rows take a forward-only lm_head pass inside the step)")
    ap.add_argument("--trace-markers", action="store_true",
                    help="profiling runs: one marker launch (gather_rows_kernel on a grid of 1237 workgroups) behind the fence that opens the timed "
                         "region and one behind the fence that closes it, so tools/kernel_stats_steps.py can cut the kernel trace to the timed steps")
    ap.add_argument("--same-batch", action="store_true", help="A/B aid: replay one batch (splice-plan cache hit) instead of a fresh one per step")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even with one rank (test hook)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                              # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.force_dist:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product path has no CPU fallback")
    # test hook: VP_TEST_SHARED_GPU=1 puts every rank on GPU 0 with the gloo backend (RCCL refuses two ranks on one device), so the
    # N>1 control flow of this script can be exercised on a one-GPU box; the real runs are one process per GPU over RCCL
    shared = os.environ.get("VP_TEST_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from visper_lm_amd import ops
    from visper_lm_amd.config import llama3_8b, llama3_8b_convnext, phi3_mini
    from visper_lm_amd.engine import Engine

    step_tf = STEP_TF_PER_IMAGE
    if args.workload == "convnext":
        cfg, step_tf = llama3_8b_convnext(), 69.07
    elif args.workload == "phi3":
        cfg, step_tf = phi3_mini(), 71.94
        cfg.image_depth = dict(cfg.image_depth); cfg.image_gen = dict(cfg.image_gen); cfg.image_seg = dict(cfg.image_seg)
        if args.text_len == 1449:
            args.text_len, args.batch = 3497, min(args.batch, 4)      # post-splice S = 4096
    elif args.workload == "ift":
        # IFT-stage step on the same shapes (NTP only, whole Llama-3-8B trainable: + 238 TFLOP of weight gradients,
        # 8 B-parameter AdamW, per-layer gradient buckets); not the headline metric
        cfg, step_tf = llama3_8b(aux_mode="", num_task_tokens=0, train_llm=True), 95.5
        if args.text_len == 1449:
            args.text_len = 1473                                          # post-splice S = T - 1 + 576 = 2048 without task tokens
    elif args.workload == "pt6":
        # the reference's own PT recipe (scripts/train/pretrain.sh:19-23): two layers per task.  Per image on top of configs[1]'s 65.84 TF:
        # three more heads fwd+bwd = 3 x (0.018 + 0.044 + 0.146) TF (BASELINE.md section 2 head rows, x3 for the step)
        cfg, step_tf = llama3_8b(), STEP_TF_PER_IMAGE + 3 * (0.018 + 0.044 + 0.146)
        cfg.image_depth = dict(cfg.image_depth, depth_layer_indices="18-20")
        cfg.image_seg = dict(cfg.image_seg, seg_layer_indices="10-18")
        cfg.image_gen = dict(cfg.image_gen, img_layer_indices="12-20")
    else:
        cfg = llama3_8b()
    cfg.depth_decoder = not args.no_depth_decoder and args.workload != "ift"
    if args.layers:
        cfg.num_hidden_layers = args.layers
        cfg.image_gen["img_layer_indices"] = str(min(20, args.layers))
        cfg.image_depth["depth_layer_indices"] = str(min(18, args.layers))
        cfg.image_seg["seg_layer_indices"] = str(min(18, args.layers))
    model = None
    if args.api == "model":
        # the reference-named module owns the parameters; its engine is built from the module's state_dict (EngineModule._get_engine)
        from visper_lm_amd import model as M_
        if args.workload == "ift":
            mcls, ccls = M_.LlavaLlamaForCausalLM, M_.LlavaConfig
        elif args.workload == "phi3":
            mcls, ccls = M_.OlaLlavaPhi3ForCausalLM, M_.OlaLlavaPhi3Config
        else:
            mcls, ccls = M_.OlaLlavaLlamaForCausalLM, M_.OlaLlavaLlamaConfig
        cd = cfg.to_dict()
        for k in getattr(ccls, "STORED_KEYS_IGNORED", ()):
            cd.pop(k, None)
        mcfg = ccls(**cd)
        mcfg.reference_outputs = bool(args.reference_outputs)
        model = mcls(mcfg, device=dev, init="random", seed=0)
        eng = model._get_engine()
        cfg = eng.cfg
    else:
        eng = Engine(cfg, device=dev)
    legs = ["torch"]
    if world > 1:
        legs = {"both": ["torch", "native"], "torch": ["torch"], "native": ["native"]}[args.transports]
        if os.environ.get("VP_COMM") in ("torch", "native") and args.transports == "both":      # VP_COMM names the primary transport
            legs = [os.environ["VP_COMM"]] + [t for t in ("torch", "native") if t != os.environ["VP_COMM"]]
        if shared:
            legs = ["torch"]                            # the one-GPU test hook runs over gloo: no RCCL communicator to build
    eng.set_distributed(rank, world, transport=legs[0])
    if model is None:
        eng.keep_logits = eng.keep_states = bool(args.reference_outputs)
        eng.init_random(seed=0)                   # identical weights on every rank
    # A FRESH batch every step, as a dataloader delivers it (ola_vlm_train.py:882-925): new input_ids / labels each step, so the host
    # splice plan (ola_arch.py:256-444 restated in splice.host_plan), its H2D copy and the head tables are paid inside the timed
    # region; images / teacher targets rotate through 4 distinct sets already resident in HBM.
    n_total = args.warmup + args.steps
    pool = [make_batch(cfg, args.batch, args.text_len, rank + 1000 * j, dev) for j in range(4)]
    gi = torch.Generator().manual_seed(4321 + rank)
    fresh = []
    for j in range(n_total):
        b = dict(pool[j % len(pool)])
        ids = torch.randint(0, 1000, (args.batch, args.text_len), generator=gi)
        ids[:, cfg.num_sys_tokens] = -200
        lab = ids.clone()
        lab[:, :cfg.num_sys_tokens + 7] = -100
        b["input_ids"], b["labels"] = ids, lab
        b["images_resident"] = True                 # the image / target pool above was written to HBM before the first step was enqueued
        fresh.append(b)

    torch.cuda.synchronize(dev)                      # (images_resident: the pool is in HBM before any step is enqueued)
    from visper_lm_amd import optim
    total_steps = 2181                                  # LLaVA-558K / global batch 256 (scripts/train/pretrain.sh), 3 % warm-up, cosine
    n_warm = optim.warmup_steps(total_steps, 0.03)
    it = [0]

    def step():
        b = fresh[it[0] % n_total] if not args.same_batch else fresh[0]
        lr_mult = optim.cosine_with_warmup(it[0], total_steps, n_warm)
        if model is None:
            out = eng.train_step(b)
            eng.optimizer_step(lr=args.lr, lr_mult=lr_mult)                  # wd 0, no clipping: pretrain.sh
        else:
            # what the reference's trainer does per micro-batch (HF Trainer.training_step under ola_vlm_train.py:1297-1327): forward through
            # the model class, backward through autograd, optimizer, zero_grad
            res = model(**b, output_logits=True) if args.read_logits else model(**b)     # (output_logits=True: computed inside the step, not lazily)
            if args.read_logits:
                _lg = res.logits                                   # materialise the reference's fp32 [B, S, V] logits (lazy otherwise)
                assert _lg.dtype == torch.float32 and _lg.shape[-1] == cfg.vocab_size
                del _lg
            res.loss.backward()
            model.optimizer_step(lr=args.lr, lr_mult=lr_mult)
            model.zero_grad(set_to_none=True)
            out = model._last
            out["loss"] = res.loss.detach()
        it[0] += 1
        return out

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def marker():
        n = 1237 * 4
        src = torch.zeros(8, 8, device=dev, dtype=torch.bfloat16)
        ops.gather_rows([src], torch.zeros(n, device=dev, dtype=torch.int32), torch.zeros(n, device=dev, dtype=torch.int32), 8,
                        torch.empty(n, 8, device=dev, dtype=torch.bfloat16))

    def timed_leg(n_warm_, n_steps, profile=False):
        """W untimed steps, then exactly K timed steps between barrier + device-synchronize fences; returns (seconds of THIS rank, last out,
        per-GEMM HIP-event records)."""
        out_ = None
        for _ in range(n_warm_):
            out_ = step()
        fence()
        if profile:
            ops.GEMM_PROF = []
        if args.trace_markers and profile:
            marker()
        t0_ = time.perf_counter()
        for _ in range(n_steps):
            out_ = step()
        fence()
        el = time.perf_counter() - t0_
        if args.trace_markers and profile:
            marker()
            torch.cuda.synchronize()
        prof_ = None
        if profile:
            prof_, ops.GEMM_PROF = ops.GEMM_PROF, None
        return el, out_, prof_

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    dt_own, out, prof = timed_leg(args.warmup, args.steps, profile=True)
    dt = max_over_ranks(dt_own)
    diag = None
    if dist is not None:
        diag = multi_gpu_report(args, eng, dist, dev, rank, world, legs, dt_own, fresh, step, timed_leg, max_over_ranks, fence, shared)
    S = out["plan"]["S"]
    n_valid_rows = out["plan"]["n_valid"]
    loss = float(out["loss"])
    if loss != loss or abs(loss) == float("inf"):
        raise SystemExit(f"bench.py: non-finite loss ({loss}) after the timed steps — the measurement is void")
    ms = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt

    # roofline of the dominant kernel: the persistent 256x256x64 bf16 MFMA GEMM (every GEMM with >= 192 256x256 output tiles routes to it:
    # gemm_nt_256w4 since round 3, gemm_nt_256p8 with VP_GEMM_W4=0 or for the few shapes the 4-wave kernel does not take; ~80 % of the step).
    # achieved = algorithmic 2*M*N*K of those launches / their HIP-event-timed duration on the launch stream.
    w4 = os.environ.get("VP_GEMM_W4", "1") != "0"

    def is_dom(shp, kind):
        # the launcher's routing (gemm.hip vp_gemm_bf16 / vp_gemm_bf16_swiglu): >= 192 256x256 tiles; the 4-wave kernel takes the lean launches
        # (no bias / activation) of 256/256/128-aligned problems, the 8-phase kernel the rest -> ONE kernel's launches, comparable with its
        # rocprofv3 --stats line (profiles/rNN_bench_kernel_stats.csv)
        M_, N_, K_ = shp
        if kind == "tn" or not (K_ % 64 == 0 and M_ >= 256 and N_ >= 256 and ((M_ + 255) // 256) * ((N_ + 255) // 256) >= 192):
            return False
        lean = kind == "nt_lean" and M_ % 256 == 0 and N_ % 256 == 0 and K_ % 128 == 0
        return lean if w4 else True
    big = [(e0.elapsed_time(e1), f) for e0, e1, f, shp, kind in prof if is_dom(shp, kind)]
    # the same K loop under its other two epilogue instantiations (rocprofv3 lists them as gemm_nt_256w4<false, 1> / <false, 2>): the QKV projection
    # with RoPE (+ row scale) and the RMSNorm-fold launches (residual + sums of squares, fused SwiGLU forward + row scale)
    fam = [(e0.elapsed_time(e1), f) for e0, e1, f, shp, kind in prof
           if is_dom(shp, kind) or (w4 and kind in ("nt_fold", "nt_rope") and is_dom(shp, "nt_lean"))]
    g_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in prof)
    g_fl = sum(f for _, _, f, _, _ in prof)
    b_ms, b_fl = sum(t for t, _ in big), sum(f for _, f in big)
    achieved = b_fl / (b_ms * 1e-3) / 1e12 if b_ms > 0 else 0.0
    traffic, latest = None, None                       # HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/)
    try:
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        latest = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_w4.json" if w4 else "_pmc_p8.json"))[-1]        # newest round's PMC pass of the kernel that runs
        with open(os.path.join(pdir, latest)) as fh:
            traffic = json.load(fh).get("hbm_bytes_per_launch") if args.workload == "llama3_8b" else None
    except (OSError, ValueError):
        pass
    roof = {"bound": "mfma", "kernel": ("gemm_nt_256w4 (bf16 MFMA 16x16x32, persistent 256x256x64 tiles, one wave per SIMD with a 128x128 sub-tile in AGPRs, "
                                        "hand-scheduled K loop: one LDS-DMA piece / ds_read per MFMA gap)" if w4 else
                                        "gemm_nt_256p8 (bf16 MFMA 16x16x32, persistent 256x256x64 tiles, two wave groups in ping-pong, 4 phases per K-tile)"),
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TF, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TF, 4),
            "traffic": traffic, "traffic_source": (f"profiles/{latest} (a separate rocprofv3 --pmc pass of this command, committed; not measured in this run)"
                                                   if traffic is not None else None), "launches_per_step": len(big) // max(args.steps, 1),
            "avg_launch_ms": round(b_ms / max(len(big), 1), 4), "tflop_per_launch": round(b_fl / max(len(big), 1) / 1e12, 4),
            "kernel_ms_per_step": round(b_ms / args.steps, 2), "all_gemm_ms_per_step": round(g_ms / args.steps, 2),
            "all_gemm_tflop_per_step": round(g_fl / args.steps / 1e12, 2),
            "kernel_instance": "gemm_nt_256w4<false, 0> (the lean epilogues; its rocprofv3 --stats line)" if w4 else "gemm_nt_256p8",
            # all three epilogue instantiations of the same K loop (<false, 0 | 1 | 2>: + QKV with RoPE, + the RMSNorm-fold launches)
            "family": {"launches_per_step": len(fam) // max(args.steps, 1), "kernel_ms_per_step": round(sum(t for t, _ in fam) / args.steps, 2),
                       "achieved": round(sum(f for _, f in fam) / max(sum(t for t, _ in fam), 1e-9) / 1e9, 1),
                       "frac": round(sum(f for _, f in fam) / max(sum(t for t, _ in fam), 1e-9) / 1e9 / PEAK_BF16_TF, 4)},
            "schedule": "heads + DPT decoder and (images resident) the frozen tower run on a side stream beside these launches: their HIP-event "
                        "durations include whatever CU-time the side stream took"}
    # The same launches with the step on ONE stream (3 extra steps outside the timed region, world 1 only): what the kernel does when nothing shares
    # the chip with it — the figure the earlier rounds' `frac` was.
    if dist is None and not args.no_probes and os.environ.get("VP_HEADS_STREAM", "1") != "0":
        keep = {k: os.environ.get(k) for k in ("VP_HEADS_STREAM", "VP_TOWER_STREAM")}
        os.environ["VP_HEADS_STREAM"], os.environ["VP_TOWER_STREAM"] = "0", "0"
        try:
            el_s, _, prof_s = timed_leg(1, 3, profile=True)
            big_s = [(e0.elapsed_time(e1), f) for e0, e1, f, shp, kind in prof_s if is_dom(shp, kind)]
            ms_s, fl_s = sum(t for t, _ in big_s), sum(f for _, f in big_s)
            roof["one_stream_schedule"] = {"ms_per_step": round(el_s / 3 * 1e3, 2), "avg_launch_ms": round(ms_s / max(len(big_s), 1), 4),
                                           "achieved": round(fl_s / max(ms_s, 1e-9) / 1e9, 1), "frac": round(fl_s / max(ms_s, 1e-9) / 1e9 / PEAK_BF16_TF, 4),
                                           "what": "VP_HEADS_STREAM=0 VP_TOWER_STREAM=0, 3 steps after the timed region"}
        except Exception as e:                       # (a measurement aid must not take the line down)
            roof["one_stream_schedule"] = {"error": repr(e)[:200]}
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    # whole-step fraction of the bf16 MFMA peak, priced on the FLOPs this rank EXECUTED (every GEMM launch's 2MNK as recorded live — the
    # lm_head GEMMs only cover the labelled rows — plus the causal decoder attention at S^2/2, backward 2x forward, and the ViT attention);
    # the nominal BASELINE.md table figure (lm_head over every row) is kept beside it
    hd_all = cfg.num_attention_heads * cfg.head_dim
    attn_tf = args.batch * (3 * 4.0 * S * S / 2 * hd_all * cfg.num_hidden_layers +
                            (0 if cfg.is_convnext else 4.0 * 577 * 577 * cfg.vit_hidden * (cfg.vit_layers - 1))) / 1e12
    exec_tf = g_fl / args.steps / 1e12 + attn_tf
    roof["executed_tflop_per_step"] = round(exec_tf, 2)

    def assemble():
        roof["step_frac_of_peak"] = round(exec_tf / (ms * 1e-3) / PEAK_BF16_TF, 4)           # of the headline leg's step time
        roof["step_frac_of_peak_nominal_table"] = round(value / world * step_tf / PEAK_BF16_TF, 4)
        return {"metric": "train-step images/sec (NTP+distill), ViT-L+Llama3-8B seq2048" if args.workload == "llama3_8b" else
               ("train-step images/sec (NTP only, IFT stage: whole LLM trainable), ViT-L+Llama3-8B seq2048" if args.workload == "ift"
                else f"train-step images/sec (NTP+distill), {args.workload}"), "value": round(value, 4), "unit": "images/s",
               "n_gpus": (dist.get_world_size() if dist is not None else 1), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, random images/tokens/targets)",
               "config": {"workload": {"llama3_8b": "configs[1]: CLIP-ViT-L/14-336 + Llama-3-8B PT step, 3 distill heads (d18,s18,g20), 336px",
                                       "convnext": "configs[3]: CLIP-ConvNeXt-XXL (768px) + Llama-3-8B PT step, 3 distill heads",
                                       "phi3": "configs[4]: CLIP-ViT-L/14-336 + Phi-3-mini PT step, 3 distill heads, seq 4096",
                                       "ift": "SURVEY f-2: CLIP-ViT-L/14-336 + Llama-3-8B IFT step (NTP only, whole LLM trainable)",
                                       "pt6": "the reference's script-default PT recipe: configs[1] with 6 distill heads d18-20_s10-18_g12-20 (pretrain.sh:20)"}[args.workload],
                          "global_batch": args.batch * world, "per_gpu_batch": args.batch, "seq_len": S, "text_len": args.text_len,
                          "parallelism": f"dp{world}", "decoder_layers": cfg.num_hidden_layers, "loss": round(loss, 4),
                          "depth_decoder": bool(cfg.depth_decoder), "fresh_batch_per_step": not args.same_batch,
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                          "lm_head_rows": f"{n_valid_rows} labelled of {args.batch * S} (rows with label -100 skip lm_head + CE: zero loss, zero d_logits)",
                          "api": ("model: out = %s(**batch); out.loss.backward(); model.optimizer_step(lr); model.zero_grad() "
                                  "(the drop-in boundary: ola_vlm_train.py:1297-1327 -> ola_llama.py:190-244)" % type(model).__name__) if model is not None
                                 else "engine: Engine.train_step(batch) + Engine.optimizer_step(lr) called directly",
                          "outputs": ("reference (all L + 1 layer states returned every step, fp32 logits of all B x S rows %s: the model class's contract "
                                      "default config.reference_outputs=True)" % ("READ every step" if args.read_logits else "handed out lazily, not read")
                                      if args.reference_outputs
                                      else "lean: loss + per-layer losses + embeddings (config.reference_outputs=False: no logits tensor, lm_head + CE on "
                                           "labelled rows only; the contract-default mode is timed under extras.reference_outputs)"),
                          "valid": args.layers is None},
               "roofline": roof, **({"multi_gpu": diag} if diag is not None else {})}

    if dist is not None and len(legs) > 1:
        alt_transport_legs(args, eng, dist, rank, world, legs, diag, timed_leg, max_over_ranks,
                           emit_partial=lambda: print(json.dumps(assemble(), default=str), flush=True))
    if diag is not None:
        diag.pop("_ctx", None)
        # Headline leg (north_star: "RCCL all-reduce ... on a side HIP stream" = the C ABI's own communicator, vp_comm_*): the torch.distributed
        # leg runs FIRST because it cannot take the job down, the native leg second behind the watchdog; when the native leg completed — same
        # K steps between the same fences, gradients and gathered targets bit-identical on every rank (asserted in exchange_points) — ITS time is
        # the line's `value`; otherwise the torch leg's stays.  Every rank takes the same decision from the same all-reduced numbers.
        nat = diag.get("native") if legs[0] != "native" else None
        diag["headline_transport"] = legs[0]
        if isinstance(nat, dict) and "ms_per_step" in nat and "error" not in nat:
            ms = nat["ms_per_step"]
            dt = ms * 1e-3 * args.steps
            value = args.batch * world * args.steps / dt
            diag["headline_transport"] = "native"
        info = diag.get(diag["headline_transport"], {}).get("vp_comm_info")
        diag["n_ranks_seen"] = {"torch.distributed": dist.get_world_size(), "vp_comm_info": info["world"] if info else None}
    if rank == 0:
        res = assemble()
        # the stand-alone probes run AFTER the timed region and must never cost the line: a failing probe is reported as its error string
        if args.workload in ("llama3_8b", "convnext", "phi3", "pt6") and not args.no_probes:
            try:
                roof["k11"] = {"what": "distillation-loss reduction vp_emb_loss_fwd/bwd alone: 25 back-to-back launches in a HIP graph, replayed (N > 1: a python loop), HIP events on its stream, per launch (launch gap included)",
                               "peak_GBps": 8000.0, "world1": k11_probe(cfg, args.batch, 1, dev), "world8_shaped": k11_probe(cfg, args.batch, 8, dev)}
            except Exception as e:                      # noqa: BLE001
                roof["k11"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # the chip clocks to its 1400 W package cap: the 2.5 PFLOP/s peak assumes 2.4 GHz; report the clock the kernel actually sustains
        if not args.no_probes:
            try:
                ck = clock_probe(dev)
                roof["clock"] = ck
                roof["frac_at_sustained_clock"] = round(achieved / (PEAK_BF16_TF * ck["shader_clock_mhz"] / ck["nominal_mhz"]), 4)
            except Exception as e:                      # noqa: BLE001
                roof["clock"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and args.workload == "llama3_8b" and (not args.no_extras or not args.no_cpu_baseline):
            del eng, fresh, pool, model, out
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            if not args.no_extras and args.layers is None:
                res["extras"] = run_extras(args, res["ms_per_step"])
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(cfg, dev)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
