#!/bin/bash
# Kernel experiments: build a variant of the library with ONE source compiled under extra flags, next to the product
# library (csrc/variants/lib_<name>.so, git-ignored; travels to the GPU box).  Tools take it through GDMAE_LIB.
#   tools/build_variant.sh <name> <file.hip> [extra hipcc flags...]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../gd-mae_amd/csrc"
mkdir -p variants
extra=""
case $src in attention_t32.hip|attention_t16.hip|vfe_fused.hip|vfe_layer2.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result $extra "$@" -c $src -o variants/${name}_${src%.hip}.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_${name}.so $objs variants/${name}_${src%.hip}.o -L/opt/rocm/lib -lhipblaslt
echo built variants/lib_${name}.so
