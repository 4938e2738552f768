#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
./build_tmp/probe_mfma_layout > gpurun_out/r5d_mfma_layout.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_anchor.py tests/test_gpu_soak_cases.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5d_tests.txt
cat gpurun_out/r5d_tests.txt
for i in 1 2 3; do
  for L in gmmloc_amd/variants/lib_noallsolve.so gmmloc_amd/libgmmloc_hip.so; do
    GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|$L plain |"
  done
done 2>&1 | tee gpurun_out/r5d_ab_allsolve.txt
for L in gmmloc_amd/variants/lib_noallsolve.so gmmloc_amd/libgmmloc_hip.so; do
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 1 2>/dev/null | tail -1 | sed "s|^|$L prior |"
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 1000 2>/dev/null | tail -1 | sed "s|^|$L plain1000 |"
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 8192 3 0 400 2>/dev/null | tail -1 | sed "s|^|$L plain400 |"
done 2>&1 | tee -a gpurun_out/r5d_ab_allsolve.txt
head -20 gpurun_out/r5d_mfma_layout.txt
