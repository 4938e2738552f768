#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## tests"
python -m pytest tests/test_gpu_pose.py tests/test_gpu_chain.py tests/test_gpu_adapter.py tests/test_gpu_threads.py -q -x 2>&1 | tail -5
echo "## phase profile (new arithmetic)"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 1000 1 2>/dev/null
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 1200 1 2>/dev/null
echo "## latency, before"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_posebefore.so python tools/latency.py 2>/dev/null | grep optimizeCurrentPose
echo "## latency, after"
python tools/latency.py 2>/dev/null | grep optimizeCurrentPose
echo "## chain, before / after"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_posebefore.so python tools/chain_time.py 2>/dev/null | cut -c1-700
python tools/chain_time.py 2>/dev/null | cut -c1-700
} > gpurun_out/r5_pose_ab.txt 2>&1
cat gpurun_out/r5_pose_ab.txt
