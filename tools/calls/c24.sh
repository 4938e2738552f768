#!/bin/bash
# matcher: the queries that walk again, listed (dense) - tests, soak, stations, A/B against the library of the commit before
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c24.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_match.py tests/test_gpu_chain.py tests/test_gpu_pose.py tests/test_gpu_adapter.py -x -q -m gpu 2>&1 | tail -3 >> $O
timeout 900 python tools/soak_match.py 600 2>&1 | tail -4 >> $O
for np in 1500 3000; do for b in 1 2048; do
  MATCH_LEGS_NP=$np GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --prof 2>/dev/null | grep leg >> $O
done; done
for r in 1 2; do for b in 1 256 2048; do
  echo "== prefuse B=$b" >> $O
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prefuse.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --reps 20 2>/dev/null | grep leg >> $O
  echo "== listed B=$b" >> $O
  timeout 300 python tools/match_legs.py --legs proj,frame --B $b --reps 20 2>/dev/null | grep leg >> $O
done; done
for r in 1 2; do
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prefuse.so timeout 600 python tools/chain_ab.py prefuse 2>/dev/null | grep label >> $O
  timeout 600 python tools/chain_ab.py fused_listed 2>/dev/null | grep label >> $O
done
cat $O
