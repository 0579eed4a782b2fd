"""Per-kernel statistics of the TIMED STEPS only (rocprofv3's own --stats summary mixes Engine.init_random's launches — 481 normal_,
129 of the 325 transposes, ... — with the steps, and summing it gave 536 ms "per step").  bench.py --trace-markers launches a marker kernel
(gather_rows_kernel on a grid of 1237 workgroups: no launch of the step has that grid) right behind the fence that opens the timed region and
right behind the one that closes it; this script keeps the dispatches of the kernel trace between the two markers.

    python tools/kernel_stats_steps.py <kernel_trace.csv> <steps> [out.csv]
"""
import collections
import csv
import sys

MARK_WG = 1237


def main(path, steps, out=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "gather_rows_kernel" in r["Kernel_Name"] and
             int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) in (MARK_WG, MARK_WG * 256)]
    if len(marks) < 2:
        raise SystemExit(f"expected two marker dispatches, found {len(marks)} (run bench.py with --trace-markers)")
    a, b = marks[-2], marks[-1]
    sel = rows[a + 1:b]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0, 10 ** 18, 0])
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        x = agg[r["Kernel_Name"]]
        x[0] += 1; x[1] += d; x[2] = min(x[2], d); x[3] = max(x[3], d)
    tot = sum(x[1] for x in agg.values())
    lines = ["Name,Calls,CallsPerStep,TotalDurationNs,MsPerStep,AverageNs,MinNs,MaxNs,PercentOfKernelTime"]
    for k, (n, s, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{k}\",{n},{n / steps:.1f},{s},{s / steps / 1e6:.3f},{s / n:.0f},{lo},{hi},{100.0 * s / tot:.2f}")
    head = (f"# timed steps only: {len(sel)} dispatches between the two markers, {steps} steps, wall between the markers {(t1 - t0) / 1e6:.2f} ms "
            f"= {(t1 - t0) / steps / 1e6:.2f} ms/step (under the profiler); summed kernel durations {tot / steps / 1e6:.2f} ms/step "
            f"(two streams overlap: the sum exceeds the wall time)")
    text = head + "\n" + "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(head)
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None)
