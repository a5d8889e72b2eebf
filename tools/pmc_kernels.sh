#!/bin/bash
# PMC counters of selected kernels over a short bench run (separate rocprofv3 --pmc passes, no trace domains).
# Usage: bash tools/pmc_kernels.sh <out.csv> <kernel-name-regex> "<counters pass 1>" ["<counters pass 2>" ...]
set -u
OUT=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
echo "kernel,counter,dispatches,sum,avg" > $OUT
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C -f csv -T -d /tmp/pmc_$i -- python /root/repo/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-also > /tmp/pmc_$i.log 2>&1
  F=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$F" "$PAT" >> $OUT <<'PY'
import csv, sys, re, collections
f, pat = sys.argv[1], re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
    if not pat.search(name):
        continue
    a = agg[(name, r["Counter_Name"])]
    a[0].add(r["Dispatch_Id"])
    a[1] += float(r["Counter_Value"])
for (k, c), (ds, v) in sorted(agg.items()):
    print(f"{k},{c},{len(ds)},{v:.0f},{v / max(len(ds), 1):.1f}")
PY
done
