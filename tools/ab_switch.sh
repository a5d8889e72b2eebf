#!/bin/bash
# same-box A/B of one environment switch on the bench value at 8 and 4 frames per step, alternating: tools/ab_switch.sh VAR
V=$1
for b in 8 4; do
  for r in 1 0 1 0; do
    echo -n "frames $b $V=$r: "
    env $V=$r python /root/repo/bench.py --steps 60 --warmup 10 --batch-per-gpu $b --no-cpu-baseline --no-roofline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
