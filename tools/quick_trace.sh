#!/bin/bash
# Kernel stats + one-step trace of the default bench (no PMC).  Usage (via gpurun): bash tools/quick_trace.sh <tag> [bench args]
set -u
TAG=${1:-t}; shift
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -f csv -T -d /tmp/prof_kt -- python /root/repo/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-also "$@" > $OUT/kt.log 2>&1
python /root/repo/tools/trace_step.py $(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1) $OUT/step_trace.csv
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
tail -2 $OUT/kt.log | cut -c1-400
python - $OUT/step_trace.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["name"].split("(")[0].replace("void ", "")[:60]
    agg[n][0] += 1; agg[n][1] += float(r["dur_us"])
tot = sum(v[1] for v in agg.values())
print("launches", len(rows), "kernel us", round(tot, 1))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f %4d  %s" % (v[1], v[0], k))
PY
