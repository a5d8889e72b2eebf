#!/bin/bash
# same-box A/B of variant libraries (tools/build_variant.sh): tools/ab_variants.sh base <name> <name> ... (base = the product library)
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset GDMAE_LIB; else export GDMAE_LIB=gd-mae_amd/csrc/variants/lib_$v.so; fi
  echo "== $v"; python tools/phase_times.py ${FRAMES:+--frames $FRAMES} 2>&1 | grep "wall/step"
done; done
