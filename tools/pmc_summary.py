"""Per-kernel averages of rocprofv3 --pmc counters.
   python tools/pmc_summary.py <dir or *_counter_collection.csv> [substring filter]
Works on the csv output (--output-format csv): rows = (dispatch, counter)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name") or r.get("kernel_name")
            c = r.get("Counter_Name") or r.get("counter_name")
            v = float(r.get("Counter_Value") or r.get("counter_value") or 0)
            a = acc[k][c]
            a[0] += v; a[1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = []
for k, cs in acc.items():
    if flt and flt not in k:
        continue
    out.append({"kernel": k, "launches": max(v[1] for v in cs.values()), **{c + "_avg": v[0] / v[1] for c, v in cs.items()}})
out.sort(key=lambda r: -r["launches"])
print(json.dumps(out, indent=1))
