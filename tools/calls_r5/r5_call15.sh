#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_anchor.py tests/test_gpu_soak_cases.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r5o_tests.txt
for i in 1 2; do
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_persist.so python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|persist |"
  python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|persist+diag |"
done 2>&1 | tee gpurun_out/r5o_ab_lambda.txt
for a in "4096 3 0 1000" "4096 3 1" "16384 3 0"; do
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_persist.so python tools/refine_only.py $a 2>/dev/null | tail -1 | sed "s|^|persist $a: |"
python tools/refine_only.py $a 2>/dev/null | tail -1 | sed "s|^|persist+diag $a: |"
done | tee -a gpurun_out/r5o_ab_lambda.txt
