"""Per-class kernel time of one training step (hand-written HIP / hipBLASLt / MIOpen / framework) from the compact step trace
written by tools/trace_step.py.  Usage: class_shares.py <step_trace.csv> <out.json> [source label]"""
import collections
import csv
import json
import sys

src, dst = sys.argv[1:3]
label = sys.argv[3] if len(sys.argv) > 3 else src
rows = list(csv.DictReader(open(src)))


def klass(name):
    if name.startswith(("Cijk_", "Custom_Cijk_")):
        return "hipBLASLt"
    if name.startswith(("k_", "gd_scan", "void k_", "void gd_scan")) or "k_" in name[:6]:
        return "hand_written_hip"
    if "miopen" in name.lower() or name.startswith(("naive_conv", "Im2", "gridwise", "igemm")):
        return "MIOpen"
    return "ATen"


shares = collections.defaultdict(lambda: {"launches": 0, "us": 0.0})
top = collections.defaultdict(lambda: [0, 0.0])
queues = collections.Counter(r.get("queue", "") for r in rows)
for r in rows:
    n = r["name"].replace("void ", "").split("(")[0].split("<")[0]
    k = klass(n)
    d = float(r["dur_us"])
    shares[k]["launches"] += 1
    shares[k]["us"] += d
    top[n[:60]][0] += 1
    top[n[:60]][1] += d
tot = sum(v["us"] for v in shares.values())
t0 = float(rows[0]["start_us"])
span = float(rows[-1]["start_us"]) + float(rows[-1]["dur_us"]) - t0
out = {"source": label, "launches": len(rows), "kernel_time_us": round(tot, 1), "step_span_us": round(span, 1),
       "launches_by_queue": dict(queues),
       "shares": {k: {"launches": v["launches"], "us": round(v["us"], 1), "frac": round(v["us"] / tot, 4)} for k, v in sorted(shares.items())},
       "top": [{"kernel": k, "launches": v[0], "us": round(v[1], 1)} for k, v in sorted(top.items(), key=lambda kv: -kv[1][1])[:12]]}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["shares"]), out["launches"], out["kernel_time_us"])
