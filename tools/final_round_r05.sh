# Round-5 final evidence, one box, in this order (run through gpurun from the repo root).
mkdir -p gpurun_out/final
python bench.py --steps 10 --warmup 3 > gpurun_out/final/bench_default.log 2>&1; tail -1 gpurun_out/final/bench_default.log > gpurun_out/final/r05_bench.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probes 2>/dev/null | tail -1 > gpurun_out/final/r05_bench_plain.json
for w in phi3 convnext ift pt6; do python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/r05_bench_$w.json; done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-probes --reference-outputs 2>/dev/null | tail -1 > gpurun_out/final/r05_bench_reference_outputs.json
bash tools/profile_round.sh r05f > gpurun_out/final/profile_round.log 2>&1
bash tools/pmc_round.sh r05f > gpurun_out/final/pmc_round.log 2>&1
bash tools/attn_bwd64_clock.sh > gpurun_out/final/attn_clock.log 2>&1
bash tools/attn_kernel_times.sh > gpurun_out/final/attn_kernel_times.log 2>&1
