#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r5r_tests.txt
for P in 0 1; do
GMMLOC_MATCH_PERSIST=$P python tools/match_legs.py --legs proj,frame --reps 5 2>/dev/null | grep leg | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('persist $P', d['leg'], round(d['units_per_s']), round(d['ms_per_launch'], 4))
"
done | tee gpurun_out/r5r_match_persist.txt
