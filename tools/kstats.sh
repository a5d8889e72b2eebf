#!/bin/bash
# rocprofv3 kernel statistics of a command (no PMC): average duration per kernel name.  Usage (via gpurun): bash tools/kstats.sh <grep pattern> <command...>
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats -f csv -T -d /tmp/prof_ks -- "$@" > /tmp/prof_ks.log 2>&1
python3 - "$pat" $(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys, re
pat = re.compile(sys.argv[1])
for r in csv.DictReader(open(sys.argv[2])):
    if pat.search(r["Name"]):
        print("%-60s calls %5s avg %9.1f us total %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
