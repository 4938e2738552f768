#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for M in 2000 1001; do
python tools/refine_only.py 4096 3 0 $M 2>&1 | grep "^refine"
GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 4096 3 0 $M 2>&1 | grep "^refine"
done
python tools/refine_only.py 16384 3 0 2000 2>&1 | grep "^refine"
GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 16384 3 0 2000 2>&1 | grep "^refine"
