#!/bin/bash
# VGPR / spill / LDS figures of every kernel of one source file (cross-compiles for gfx950; no GPU needed).
# usage: tools/kernel_regs.sh rustfst_amd/csrc/sssp.hip [filter]
src=$1; filt=${2:-.}
tmp=$(mktemp -d); cd $tmp
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -c "$OLDPWD/$src" -o probe.o -save-temps=obj 2>/dev/null
python3 - "$filt" <<'PY'
import re, sys, glob, subprocess
s = open(glob.glob('*gfx950.s')[0]).read()
md = s[s.find('amdhsa.kernels'):]
for b in md.split('- .agpr_count')[1:]:
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
    name = re.sub(r'^void ', '', name).split('(')[0]
    if 'rocprim' in name or not re.search(sys.argv[1], name): continue
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size')}")
PY
rm -rf $tmp
