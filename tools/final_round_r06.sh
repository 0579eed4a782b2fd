# Round-6 final evidence, one box, in this order (run through gpurun from the repo root).
mkdir -p gpurun_out/final6
python bench.py --steps 10 --warmup 3 > gpurun_out/final6/bench_default.log 2> gpurun_out/final6/bench_default.err; tail -1 gpurun_out/final6/bench_default.log > gpurun_out/final6/r06_bench.json
for w in ift pt6; do python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/final6/r06_bench_$w.json; done
python bench.py --api engine --steps 10 --warmup 3 --no-cpu-baseline --no-probes --no-extras 2>/dev/null | tail -1 > gpurun_out/final6/r06_bench_engine_direct.json
bash tools/profile_round_r06.sh r06f 4 > gpurun_out/final6/profile_round.log 2>&1
bash tools/pmc_round.sh r06f > gpurun_out/final6/pmc_round.log 2>&1
bash tools/attn_kernel_times.sh > gpurun_out/final6/attn_kernel_times.log 2>&1
(timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/final6/gpu_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/final6/gpu_suite.log)
tail -3 gpurun_out/final6/gpu_suite.log
python - <<'PY'
import json
for n in ("r06_bench", "r06_bench_ift", "r06_bench_pt6", "r06_bench_engine_direct"):
    try:
        d = json.load(open(f"gpurun_out/final6/{n}.json")); r = d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["frac"], r["family"]["frac"], r.get("step_frac_of_peak"), {k: (v.get("ms_per_step"), v.get("delta_ms_vs_headline")) for k, v in d.get("extras", {}).items() if isinstance(v, dict)}, d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(n, "ERR", e)
PY
