#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r5s_tests.txt
python tools/match_legs.py --legs tri,bow --reps 5 2>/dev/null | grep leg | tee gpurun_out/r5s_match_legs.jsonl
timeout 900 python tools/soak_match.py 500 2>/dev/null | tail -3 | tee gpurun_out/r5s_soak_match.txt
