#!/bin/bash
# Eight ranks of bench.py on ONE GPU (VERDICT r5 task 7a): the driver's launch line with the gloo backend (RCCL refuses several ranks per
# device), config C's 4 frames per rank, every rank on cuda:0 and pinned to its eighth of the GPU's NUMA node - what eight interpreters
# issuing 4-frame steps cost each other on one host (`host_issue_ms_idle_device` = MAX over the ranks of the host time to issue one step,
# `host_ms_per_step` the same inside the loop).  The device time is shared 8 ways, so frames/s of this run mean nothing.
# usage: tools/eight_ranks_one_gpu.sh [ranks=8] [steps=8]
N=${1:-8}; STEPS=${2:-8}
cd "$(dirname "$0")/.."
export GDMAE_DIST_BACKEND=gloo GDMAE_BENCH_RESERVE_GB=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N \
  --steps $STEPS --warmup 3 --no-cpu-baseline --no-also --no-roofline 2>gpurun_out/ranks$N.err | tail -1 > gpurun_out/ranks$N.json
python - <<PY
import json
d = json.loads(open("gpurun_out/ranks$N.json").read())
gs = d.get("grad_sync", {})
print("ranks", d["n_gpus"], "frames/rank", d["config"]["frames_per_gpu"], "| ms_per_step (device shared)", d["ms_per_step"],
      "| host_ms_per_step", d["host_ms_per_step"], "| host_issue_ms_idle_device (max over ranks)", d["host_issue_ms_idle_device"],
      "| affinity", gs.get("cpu_affinity"), "| exposed_ms", gs.get("exposed_ms"), "| allocs", d.get("device_allocs_in_timed_region"))
PY
