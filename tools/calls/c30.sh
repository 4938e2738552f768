#!/bin/bash
# matcher: a listed query walked by a whole wave - tests, soak, A/B against the library of the commit before (lib_prev = d992ab4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c30.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_match.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -2 >> $O
timeout 900 python tools/soak_match.py 1000 2>&1 | tail -2 >> $O
for r in 1 2; do for b in 1 256 2048; do
  echo "== prev B=$b" >> $O
  GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prev.so timeout 300 python tools/match_legs.py --legs proj,frame --B $b --reps 20 2>/dev/null | grep leg >> $O
  echo "== wave walk B=$b" >> $O
  timeout 300 python tools/match_legs.py --legs proj,frame --B $b --reps 20 2>/dev/null | grep leg >> $O
done; done
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_prev.so timeout 600 python tools/chain_ab.py prev 2>/dev/null | grep label >> $O
timeout 600 python tools/chain_ab.py wave_walk 2>/dev/null | grep label >> $O
cat $O
