"""Print a rocprofv3 kernel_stats.csv as ms per step: tools/stats_table.py <csv> <profiled steps incl. warm-up> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot / 1e6 / steps:.1f} ms per step over {steps:.0f} steps")
for r in rows[:n]:
    print(r['Name'][:84].ljust(84), r['Calls'].rjust(6), '%8.2f ms/step' % (float(r['TotalDurationNs']) / 1e6 / steps), '%8.1f us avg' % (float(r['AverageNs']) / 1e3),
          '%5.1f%%' % (100 * float(r['TotalDurationNs']) / tot))
