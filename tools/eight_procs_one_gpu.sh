#!/bin/bash
# N independent bench.py processes (no collective) with config C's 4 frames per step sharing ONE GPU, each pinned to its N-th of the
# GPU's NUMA node: what N interpreters issuing 4-frame steps on one host cost each other (VERDICT r5 task 7a).  Reported per process:
# host_issue_ms_idle_device (host time to issue one step behind a synchronisation of its own stream) and host_ms_per_step (inside the
# loop, the wait for the plan's event excluded).  The device is time-shared, so frames/s of these runs mean nothing.
# usage: tools/eight_procs_one_gpu.sh [procs=8] [steps=20] [frames=4]
N=${1:-8}; STEPS=${2:-20}; FR=${3:-4}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export GDMAE_BENCH_RESERVE_GB=1 GDMAE_BENCH_PIN=1
pids=()
for k in $(seq 0 $((N - 1))); do
  GDMAE_BENCH_PIN_SLICE=$k/$N python bench.py --batch-per-gpu $FR --steps $STEPS --warmup 5 --no-cpu-baseline --no-also --no-roofline \
    > gpurun_out/proc${N}_$k.json 2> gpurun_out/proc${N}_$k.err &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
python - <<PY
import json, glob
rows = []
for f in sorted(glob.glob("gpurun_out/proc${N}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.append((d["host_issue_ms_idle_device"], d["host_ms_per_step"], d["ms_per_step"], d.get("pin")))
    except Exception as e:
        print(f, "FAILED", e)
print("procs $N frames/step $FR: host_issue_ms_idle_device per process", [r[0] for r in rows], "max", max(r[0] for r in rows))
print("  host_ms_per_step", [r[1] for r in rows], "| ms_per_step (device shared)", [r[2] for r in rows])
print("  pin", rows[0][3] if rows else None)
PY
