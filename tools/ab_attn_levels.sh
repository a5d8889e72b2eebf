#!/bin/bash
# the occupancy levels of an attention call one at a time (LEVELS bit mask: 1 = T16, 2 = T32, 4 = T64), product path only
for lv in 1 2 4 7; do
  for cold in "" 1; do
    echo "--- LEVELS=$lv cold=${cold:-0}"
    LEVELS=$lv COLD=$cold NOCSR=1 python /root/repo/tools/attn_layer.py 2>&1 | tail -7 | cut -c1-160
  done
done
