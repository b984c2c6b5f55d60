#!/bin/bash
# builds rustfst_amd/lib/var_<name>.so: sssp.hip compiled with extra -D flags, linked with the tree's other objects
# usage: tools/build_variant.sh NAME -DFOO [-DBAR ...]   (then WFST_LIB_PATH=rustfst_amd/lib/var_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function "$@" -c rustfst_amd/csrc/sssp.hip -o rustfst_amd/lib/var_${name}_sssp.o
objs=$(ls rustfst_amd/lib/*_hip.o rustfst_amd/lib/*_cpp.o | grep -v "lib/sssp_hip.o" | grep -v "var_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rustfst_amd/lib/var_${name}.so $objs rustfst_amd/lib/var_${name}_sssp.o
echo rustfst_amd/lib/var_${name}.so
