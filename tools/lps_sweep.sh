for n in 1000000 2000000; do
 echo "## $n default"; timeout 300 python tools/sp_repeat.py $n 30 20 2>&1 | grep -v amdgpu.ids | grep "sha1\|chain"
 for l in 4 5 6 8; do for u in 2 4; do echo "## $n lps=$l umax=$u"; WFST_SSSP_LPS=$l WFST_SSSP_UMAX=$u timeout 300 python tools/sp_repeat.py $n 30 20 2>&1 | grep "sha1\|chain"; done; done
done
