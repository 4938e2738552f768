#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_nodynb.so python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|before |"
  GMMLOC_BA_PERSIST=0 python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|persist0 |"
  GMMLOC_BA_PERSIST=1 python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|persist1 |"
done 2>&1 | tee gpurun_out/r5n_ab_persist.txt
for a in "4096 3 0 1000" "8192 3 0 400" "4096 3 1" "16384 3 0"; do
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_nodynb.so python tools/refine_only.py $a 2>/dev/null | tail -1 | sed "s|^|before $a: |"
GMMLOC_BA_PERSIST=1 python tools/refine_only.py $a 2>/dev/null | tail -1 | sed "s|^|persist1 $a: |"
done | tee -a gpurun_out/r5n_ab_persist.txt
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_anchor.py tests/test_gpu_soak_cases.py tests/test_gpu_replay.py tests/test_gpu_threads.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r5n_tests.txt
