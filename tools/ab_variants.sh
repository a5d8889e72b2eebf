mkdir -p gpurun_out/r4c
for v in base pf8 pf2 r64 r64pf8 base; do
  if [ $v = base ]; then unset GDMAE_LIB; else export GDMAE_LIB=gd-mae_amd/csrc/variants/lib_$v.so; fi
  echo "== $v"; python tools/phase_times.py 2>&1 | grep "wall/step"
done
