#!/bin/bash
# device ISA of one kernel source (same flags as the build): tools/isa.sh <file.hip> [out.s] [extra flags...]
src=$1; out=${2:-/tmp/${1%.hip}.s}; shift 2
cd "$(dirname "$0")/../gd-mae_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result "$@" -S --cuda-device-only -o $out $src 2>&1 | grep -i "error" -A3 | head -20
grep -n "private_seg_size, \|\.num_vgpr, \|vgpr_spill_count" $out | grep -v ", 0$" | head -30
