#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c26.txt; : > $O
for np in 1500 3000; do for b in 1 256; do
  MATCH_LEGS_NP=$np GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --prof 2>&1 | grep "leg\|Error\|error" >> $O
done; done
cat $O
