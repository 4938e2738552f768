#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for L in gmmloc_amd/libgmmloc_hip.so gmmloc_amd/variants/lib_asym.so gmmloc_amd/variants/lib_asymnodrop.so; do
    GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|$L |"
  done
done 2>&1 | tee gpurun_out/r5h_ab_asym.txt
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_asym.so GMMLOC_BA_SHAPE=0 timeout 600 python -m pytest tests/test_gpu_track.py -m gpu -q -x -k "matches_oracle or known_answer" 2>&1 | tail -3 | tee -a gpurun_out/r5h_ab_asym.txt
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_asymprof.so python tools/prof_ba.py 256 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5h_prof_asym.txt
