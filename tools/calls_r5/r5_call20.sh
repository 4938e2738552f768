#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_bench.sh r5a > gpurun_out/r5a_profile.log 2>&1
rm -rf gpurun_out/prof_r5a
python bench.py > gpurun_out/r5a_bench_line.json 2> gpurun_out/r5a_bench.err
tail -1 gpurun_out/r5a_bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['frac_all_points_model'], r['traffic'], r['traffic_stale'], r['hbm']['algorithmic_bytes_per_launch'], d['cpu_baseline']['value'], d['per_rank_frames_per_s'])"
head -30 gpurun_out/r5a_bench_kernel_stats.txt
