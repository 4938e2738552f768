#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_poseprof.so
{
python tools/pose_prof.py 1000 1 2>/dev/null
python tools/pose_prof.py 1200 1 2>/dev/null
python tools/pose_prof.py 1000 64 2>/dev/null
python tools/pose_prof.py 300 1 2>/dev/null
} > gpurun_out/r5_pose_prof.txt 2>&1
cat gpurun_out/r5_pose_prof.txt
