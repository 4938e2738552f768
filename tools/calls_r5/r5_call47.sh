#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## tests"
python -m pytest tests/test_gpu_pose.py tests/test_gpu_chain.py tests/test_gpu_adapter.py tests/test_gpu_threads.py tests/test_gpu_replay.py -q -x 2>&1 | tail -4
echo "## phase profile, five groups on four waves (five edges per thread)"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 1200 1 2>/dev/null
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 1100 1 2>/dev/null
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 1000 1 2>/dev/null
echo "## chain"
python tools/chain_time.py 2>/dev/null | cut -c1-900
python tools/soak_chain.py 1500 2>/dev/null | tail -2 | cut -c1-400
} > gpurun_out/r5_pose_five.txt 2>&1
cat gpurun_out/r5_pose_five.txt
