#!/bin/bash
# head groups per workgroup of the cooperative attention forward (ATTN_K64 / ATTN_K32 variants, tools/build_variant.sh): warm and cold
V=/root/repo/gd-mae_amd/csrc/variants
for v in "" k11 k21 k42; do
  for cold in "" 1; do
    echo "--- variant ${v:-product(k22)} cold=${cold:-0}"
    GDMAE_LIB=${v:+$V/lib_$v.so} COLD=$cold NOCSR= python /root/repo/tools/attn_layer.py 2>&1 | tail -7
  done
done
