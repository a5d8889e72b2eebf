#!/bin/bash
# layer tails riding with the in-projection launch (product) against launches of their own (GDMAE_LAYER_TAIL_RIDES=0): bench value at
# 8 and 4 frames per step, same box, alternating
for b in 8 4; do
  for r in 1 0 1 0; do
    echo -n "frames $b GDMAE_LAYER_TAIL_RIDES=$r: "
    GDMAE_LAYER_TAIL_RIDES=$r python /root/repo/bench.py --steps 60 --warmup 10 --batch-per-gpu $b --no-cpu-baseline --no-roofline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
