import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
ad = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
a, b = ad[-2] + 1, ad[-1]
seg = rows[a:b + 1]
from collections import Counter
c = Counter()
for i, r in enumerate(seg):
    if "at::native" in r[2] or "rocclr" in r[2]:
        c[(seg[i-1][2][:38], r[2][:70], seg[i+1][2][:38] if i+1 < len(seg) else "")] += 1
for k, v in c.most_common(25): print(v, k)
