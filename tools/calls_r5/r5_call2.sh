#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
./build_tmp/bench_mfma_point 2000 > gpurun_out/r5b_mfma_point.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5b_match_tests.txt
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 600 python tools/match_legs.py --legs proj,frame --prof > gpurun_out/r5b_match_prof.jsonl 2> gpurun_out/r5b_match_prof.err
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 600 python tools/match_legs.py --legs proj,frame --prof --B 1 --reps 20 >> gpurun_out/r5b_match_prof.jsonl 2>> gpurun_out/r5b_match_prof.err
timeout 600 python tools/match_legs.py --legs proj,frame --reps 5 > gpurun_out/r5b_match_legs.jsonl 2>&1
timeout 900 python tools/soak_match.py 300 2>/dev/null | tail -3 > gpurun_out/r5b_soak_match.txt
cat gpurun_out/r5b_mfma_point.txt gpurun_out/r5b_match_tests.txt gpurun_out/r5b_match_prof.jsonl gpurun_out/r5b_match_legs.jsonl gpurun_out/r5b_soak_match.txt
