#!/bin/bash
# same-box A/B of environment switches: tools/ab_env.sh "VAR=0" "VAR=1" ... (each argument one environment assignment list), phase_times twice
for rep in 1 2; do
for e in "$@"; do
  echo "== $e"; env $e python tools/phase_times.py ${FRAMES:+--frames $FRAMES} 2>&1 | grep "wall/step"
done; done
