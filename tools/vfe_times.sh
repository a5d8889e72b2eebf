#!/bin/bash
# per-launch kernel times of the DynVFE layers (tools/bench_vfe_layer2.py under rocprofv3): the four k_vfe1 modes in launch order
bash tools/kstats.sh "k_v2_|k_vfe1|k_bn_|k_colstats" python /root/repo/tools/bench_vfe_layer2.py 8
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_ks/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("k_vfe1")]
print("k_vfe1 launches of the last iteration (STATS, APPLY, BSTATS, DW):", ["%.1f" % x for x in v[-4:]])
PY
