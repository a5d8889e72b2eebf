#!/bin/bash
# attention experiments: both bf16 attention sources under the same extra flags -> csrc/variants/lib_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../gd-mae_amd/csrc"
mkdir -p variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $F "$@" -c attention_t32.hip -o variants/${name}_attention_t32.o &
/opt/rocm/bin/hipcc $F "$@" -c attention_t16.hip -o variants/${name}_attention_t16.o &
wait
objs=$(ls *.o | grep -v "^attention_t32.o$" | grep -v "^attention_t16.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_${name}.so $objs variants/${name}_attention_t32.o variants/${name}_attention_t16.o -L/opt/rocm/lib -lhipblaslt
echo built variants/lib_${name}.so
