#!/bin/bash
# gl_fuse_search: the record walk (float feature coordinates) against the walk from global memory (GMMLOC_FUSE_RECORDS=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_c34.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_match.py -x -q -m gpu 2>&1 | tail -3 >> $O
for r in 1 2; do for b in 1 256 2048; do
  echo "== global B=$b" >> $O
  GMMLOC_FUSE_RECORDS=0 timeout 300 python tools/match_legs.py --legs fuse --B $b --reps 20 2>/dev/null | grep leg >> $O
  echo "== records B=$b" >> $O
  timeout 300 python tools/match_legs.py --legs fuse --B $b --reps 20 2>/dev/null | grep leg >> $O
done; done
cat $O
