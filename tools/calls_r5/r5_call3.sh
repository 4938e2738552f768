#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_track.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5c_tests.txt
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 600 python tools/match_legs.py --legs proj,frame --prof > gpurun_out/r5c_match_prof.jsonl 2> gpurun_out/r5c_match_prof.err
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 600 python tools/match_legs.py --legs proj,frame --prof --B 1 --reps 20 >> gpurun_out/r5c_match_prof.jsonl 2>> gpurun_out/r5c_match_prof.err
timeout 600 python tools/match_legs.py --legs proj,frame --reps 5 > gpurun_out/r5c_match_legs.jsonl 2>&1
timeout 900 python tools/soak_match.py 300 2>/dev/null | tail -3 > gpurun_out/r5c_soak_match.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs 2> gpurun_out/r5c_bench.err | tail -1 > gpurun_out/r5c_bench.json
cat gpurun_out/r5c_tests.txt gpurun_out/r5c_match_prof.jsonl gpurun_out/r5c_match_legs.jsonl gpurun_out/r5c_soak_match.txt
python -c "
import json
d=json.loads(open('gpurun_out/r5c_bench.json').read())
r=d['roofline']
print(d['value'], d['per_rank_frames_per_s'], r['frac'], r['frac_all_points_model'], r['active_edge_share'], r['units'])
"
tail -3 gpurun_out/r5c_bench.err
