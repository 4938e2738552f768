#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
for M in 2000 1984 1700 1300 1001; do
  echo "== $M points"
  python tools/refine_only.py 4096 3 0 $M 2>&1 | grep "^refine"
  GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 4096 3 0 $M 2>&1 | grep "^refine"
done
echo "== 16384 frames x 2000"
python tools/refine_only.py 16384 3 0 2000 2>&1 | grep "^refine"
GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 16384 3 0 2000 2>&1 | grep "^refine"
} > gpurun_out/r5_two_frames.txt 2>&1
cat gpurun_out/r5_two_frames.txt
GMMLOC_BA_TWO_FRAMES=1 timeout 900 python -m pytest tests/test_gpu_track.py -x -q -m gpu 2>&1 | tail -3
