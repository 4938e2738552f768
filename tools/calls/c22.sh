#!/bin/bash
# fused chain launches against the library of the commit before (lib_prefuse), device time by events, two interleaved rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c22_chain_ab.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_pose.py -x -q -m gpu 2>&1 | tail -2 >> $O
for r in 1 2; do
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prefuse.so timeout 600 python tools/chain_ab.py prefuse 2>/dev/null | grep label >> $O
  timeout 600 python tools/chain_ab.py fused 2>/dev/null | grep label >> $O
  GMMLOC_POSE_DUAL=0 timeout 600 python tools/chain_ab.py fused_two_pose_launches 2>/dev/null | grep label >> $O
done
cat $O
