#!/bin/bash
# T = 16 forward: first two passes of a window quad planned up front (product) vs one pass at a time (variant up0 = -DATTN16_UPFRONT=0)
V=/root/repo/gd-mae_amd/csrc/variants
for lv in 1 7; do
  for v in "" up0 "" up0; do
    echo "--- LEVELS=$lv variant ${v:-product}"
    GDMAE_LIB=${v:+$V/lib_$v.so} LEVELS=$lv NOCSR= python /root/repo/tools/attn_layer.py 2>&1 | tail -1 | cut -c1-120
  done
done
