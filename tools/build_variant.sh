#!/bin/bash
# builds rustfst_amd/lib/var_<name>.so: ONE source of the library compiled with extra -D flags, linked with the tree's other
# objects: same-box A/B of kernel variants in one gpurun call (select with WFST_LIB_PATH=rustfst_amd/lib/var_NAME.so)
# usage: tools/build_variant.sh NAME [SRC=file.hip] -DFOO [-DBAR ...]     (SRC defaults to sssp.hip)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=sssp.hip
case "${1:-}" in SRC=*) src=${1#SRC=}; shift;; esac
stem=${src%.*}_${src##*.}
python -c "import sys; sys.path.insert(0, '.'); from rustfst_amd import build; build.build()"
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function "$@" -c rustfst_amd/csrc/$src -o rustfst_amd/lib/var_${name}_${stem}.o
objs=$(ls rustfst_amd/lib/*_hip.o rustfst_amd/lib/*_cpp.o | grep -v "lib/${stem}.o" | grep -v "var_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rustfst_amd/lib/var_${name}.so $objs rustfst_amd/lib/var_${name}_${stem}.o
echo rustfst_amd/lib/var_${name}.so
