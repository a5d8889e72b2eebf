#!/bin/bash
# Kernel durations (rocprofv3 kernel trace) of tools/bench_tok_gemm.py - the host loop of that tool cannot issue faster than
# ~10 us per call, so launches shorter than that must be read from the trace.  Usage (via gpurun): bash tools/trace_tok_gemm.sh <tag> [rows]
set -u
TAG=${1:-t}; ROWS=${2:-42048}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp TG_ONE=1 TG_REF=0
rm -rf /tmp/prof_tg
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_tg -- python /root/repo/tools/bench_tok_gemm.py $ROWS > $OUT/tgk_$TAG.log 2>&1
python - $(find /tmp/prof_tg -name "*kernel_stats.csv" | head -1) > $OUT/tgk_$TAG.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "k_tok_gemm" not in n and "k_tok_ffn" not in n: continue
    f = re.search(r"k_tok_ffn<(\d+)", n)
    if f:
        print("ffn    d %3s              calls %4s  avg %7.2f us  min %7.2f" % (f.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
        continue
    m = re.search(r"k_tok_gemm(_p|_multi)?<(\d+),\s*(\d+)(?:,\s*(\d+))?", n)
    if not m:
        print("?", r)
        continue
    print("%-6s K %3s N %3s epi %s  calls %4s  avg %7.2f us  min %7.2f" % (m.group(1) or "", m.group(2), m.group(3), m.group(4), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
sort $OUT/tgk_$TAG.txt -o $OUT/tgk_$TAG.txt
cat $OUT/tgk_$TAG.txt
