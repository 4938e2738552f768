#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c27.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_match.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -2 >> $O
for b in 1 2048; do
  MATCH_LEGS_NP=1500 GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --prof 2>&1 | grep "leg\|Error\|error" >> $O
done
for r in 1 2; do for b in 1 2048; do
  echo "== prefuse B=$b" >> $O
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prefuse.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --reps 20 2>/dev/null | grep leg >> $O
  echo "== listed, dealt B=$b" >> $O
  timeout 300 python tools/match_legs.py --legs proj,frame --B $b --reps 20 2>/dev/null | grep leg >> $O
done; done
timeout 600 python tools/chain_ab.py fused_listed_dealt 2>/dev/null | grep label >> $O
cat $O
