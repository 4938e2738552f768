#!/bin/bash
# k_search_by_projection without the write-only `choice` table (two frames per CU at 3 000 map points)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c35.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_match.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -2 >> $O
for np in 1500 3000 4000; do
  MATCH_LEGS_NP=$np timeout 300 python tools/match_legs.py --legs proj --B 2048 --reps 20 2>/dev/null | grep leg >> $O
done
timeout 600 python tools/chain_ab.py no_choice 2>/dev/null | grep label >> $O
cat $O
