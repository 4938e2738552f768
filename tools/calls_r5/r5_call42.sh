#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## tests"
python -m pytest tests/test_gpu_pose.py tests/test_gpu_chain.py -q -x 2>&1 | tail -3
echo "## phase profile, split shape + waves 4..7 receive the step"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 1200 1 2>/dev/null
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so python tools/pose_prof.py 2000 1 2>/dev/null
echo "## chain"
python tools/chain_time.py 2>/dev/null | cut -c1-900
python tools/latency.py 2>/dev/null | grep "optimizeCurrentPose B=1 \|optimizeCurrentPose B=4096"
} > gpurun_out/r5_pose_split2.txt 2>&1
cat gpurun_out/r5_pose_split2.txt
