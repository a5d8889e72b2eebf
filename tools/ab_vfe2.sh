# same-box kernel times of vfe_layer2 variants (tools/build_variant.sh <name> vfe_layer2.hip ...): bash tools/ab_vfe2.sh base <name> ...
for v in "$@"; do
  if [ $v = base ]; then unset GDMAE_LIB; else export GDMAE_LIB=/root/repo/gd-mae_amd/csrc/variants/lib_$v.so; fi
  echo "== $v"; bash tools/kstats.sh "k_v2_d|k_v2_max" python /root/repo/tools/bench_vfe_layer2.py 8
done
