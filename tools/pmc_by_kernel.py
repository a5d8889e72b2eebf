"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: dispatches and summed counter value.
Usage: pmc_by_kernel.py <counter_collection.csv> <COUNTER_NAME> <out.csv>"""
import collections
import csv
import sys

src, tag, dst = sys.argv[1:4]
agg = collections.defaultdict(lambda: [0, 0.0])
seen = set()
for r in csv.DictReader(open(src)):
    if r.get("Counter_Name") != tag:
        continue
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
    key = (r.get("Dispatch_Id"), r.get("Agent_Id"))
    agg[name][1] += float(r["Counter_Value"])
    if key not in seen:
        seen.add(key)
        agg[name][0] += 1
with open(dst, "w") as f:
    f.write(f"kernel,dispatches,sum_{tag},avg_{tag}\n")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{k[:80].replace(',', ';')},{n},{v:.1f},{v / max(n, 1):.1f}\n")
print("kernels:", len(agg))
