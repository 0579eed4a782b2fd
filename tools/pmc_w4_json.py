"""profiles/rNN_pmc_w4.json (the file bench.py's roofline.traffic cites) from the two rocprofv3 --pmc passes of tools/profile_round_r06.sh:
HBM (fabric) bytes per launch of the three gemm_nt_256w4 instantiations = 2 x FETCH_SIZE KB (gfx950 tallies the 128-byte requests of wide
coalesced reads at 64 bytes: MI355X_MICROARCH.md, HBM / rocprofv3 section) + WRITE_SIZE KB, averaged over the dispatches of each kernel.

    python tools/pmc_w4_json.py <fetch counter_collection.csv> <write counter_collection.csv> <tag> > profiles/<tag>_pmc_w4.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, tag):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    inst = {}
    for k in f:
        if "gemm_nt_256w4" not in k:
            continue
        n, tot = f[k]
        nw, totw = w.get(k, [0, 0.0])
        fk, wk = tot / n, (totw / nw if nw else 0.0)
        inst[k] = {"dispatches": n, "FETCH_SIZE_kb_reported_avg": round(fk, 1), "WRITE_SIZE_kb_avg": round(wk, 1),
                   "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024.0))}
    lean = next((k for k in inst if "<false, 0>" in k or "ILb0ELi0E" in k), None)
    if lean is None:
        raise SystemExit(f"no lean gemm_nt_256w4 instantiation among {list(inst)[:5]}")
    out = {"kernel": "gemm_nt_256w4<false, 0> (the lean instantiation bench.py's roofline times)", **inst[lean],
           "other_instantiations": {k: v for k, v in inst.items() if k != lean},
           "note": (f"{tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/profile_round_r06.sh), over `bench.py --steps 1 "
                    "--warmup 1 --no-cpu-baseline --no-probes --no-extras` (2 steps in the trace, the model-API path); FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B requests of wide coalesced reads at 64 B); counts L2-miss (fabric) traffic, Infinity-Cache "
                    "hits included.  Algorithmic bytes of the average lean launch (A + B + C once) are 0.6-1.2 GB: the over-fetch is the 256-tile re-read "
                    "of operand panels across XCDs, identical for hipBLASLt's kernels (profiles/r03_gemm_vs_hipblaslt_mem.txt)")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
