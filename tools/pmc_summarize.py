"""Per-kernel averages of one rocprofv3 --pmc counter (counter_collection.csv) — dev tool used by tools/profile_round.sh."""
import csv, sys, collections
path, counter = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
with open(path) as fh:
    for r in csv.DictReader(fh):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print(f"# {counter}: kernel, dispatches, total, average per dispatch")
for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k[:90]},{n},{tot:.0f},{tot / n:.1f}")
