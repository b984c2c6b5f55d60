#!/bin/bash
# same-box A/B of library variants (rustfst_amd/lib/var_<name>.so, tools/build_variant.sh): relaxation chain + level stamps
# usage: tools/ab_variants.sh OUTDIR STATES name [name ...]     ("base" = the tree's own library)
out=$1; states=$2; shift 2
mkdir -p $out
for v in "$@"; do
  if [ $v = base ]; then unset WFST_LIB_PATH; else export WFST_LIB_PATH=rustfst_amd/lib/var_$v.so; fi
  echo "== $v"
  timeout 300 python tools/sp_repeat.py $states 30 20 2>&1 | grep -v amdgpu.ids
  WFST_SSSP_RES_TRACE=$out/trace_$v.bin timeout 100 python tools/sp_repeat.py $states 3 3 > /dev/null 2>&1
  python tools/res_levels.py $out/trace_$v.bin | awk '{print $1, $2, $3, $4, $6, $7, $9, $10, $12, $13, $15, $16, $17, $18}' 
done
