#!/bin/bash
# matcher stations at the frame-at-a-time caller + chain A/B after removing the dual kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c23.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_pose.py -x -q -m gpu 2>&1 | tail -2 >> $O
for np in 1500 3000; do for b in 1 256; do
  MATCH_LEGS_NP=$np GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --prof 2>/dev/null | grep leg >> $O
done; done
for r in 1 2; do
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prefuse.so timeout 600 python tools/chain_ab.py prefuse 2>/dev/null | grep label >> $O
  timeout 600 python tools/chain_ab.py fused 2>/dev/null | grep label >> $O
done
cat $O
