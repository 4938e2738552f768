#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for L in gmmloc_amd/libgmmloc_hip.so gmmloc_amd/variants/lib_asym.so gmmloc_amd/variants/lib_asymnoprio.so; do
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|$L |"
done 2>&1 | tee gpurun_out/r5i_ab.txt
for L in baprofl asymprofl asymnopriop; do
echo "== $L"; GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_$L.so python tools/prof_ba.py 256 0 2>&1 | grep -v amdgpu.ids | grep -v "^pass [AB]: cycles\|loads " 
done | tee gpurun_out/r5i_prof.txt
