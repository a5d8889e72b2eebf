#!/bin/bash
# look-back scans of the geometry plan stand-alone: striped (product) against blocked functor evaluation (tools/build_variant.sh blocked partition.hip -DGD_SCAN_STRIPED=0)
echo "--- striped (product)"; bash /root/repo/tools/kstats.sh "scan" python /root/repo/tools/plan_standalone.py
echo "--- blocked"; GDMAE_LIB=/root/repo/gd-mae_amd/csrc/variants/lib_blocked.so bash /root/repo/tools/kstats.sh "scan" python /root/repo/tools/plan_standalone.py
