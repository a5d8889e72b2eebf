#!/bin/bash
# Round artifacts on the GPU box: full bench line, rocprofv3 kernel stats + one-step trace, PMC traffic passes.
# Usage (from the repo root, via gpurun): bash tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/profiles_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_full.log 2>&1
tail -1 $OUT/bench_full.log > $OUT/${TAG}_bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -T -d /tmp/prof_kt -- python /root/repo/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-also > /tmp/kt.log 2>&1
python /root/repo/tools/trace_step.py $(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_step_trace.csv
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -f csv -T -d /tmp/prof_$C -- python /root/repo/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-also > /tmp/pmc_$C.log 2>&1
  python /root/repo/tools/pmc_by_kernel.py $(find /tmp/prof_$C -name "*counter_collection.csv" | head -1) $C $OUT/${TAG}_pmc_${C}_by_kernel.csv
done
ls -la $OUT
