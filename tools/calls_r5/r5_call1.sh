#!/bin/bash
# round 5, first GPU call: matcher counters + prof, MFMA point micro-benchmark, batch-size sweep of the bench step
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
./build_tmp/bench_mfma_point 2000 > gpurun_out/r5a_mfma_point.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_match.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r5a_match_tests.txt
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 600 python tools/match_legs.py --legs proj,frame --prof > gpurun_out/r5a_match_prof.jsonl 2> gpurun_out/r5a_match_prof.err
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so timeout 600 python tools/match_legs.py --legs proj,frame --prof --B 1 --reps 20 >> gpurun_out/r5a_match_prof.jsonl 2>> gpurun_out/r5a_match_prof.err
timeout 1500 bash tools/pmc_match.sh r5a > gpurun_out/r5a_pmc_match.log 2>&1
for BATCH in 4096 8192 16384; do
  timeout 600 python bench.py --batch $BATCH --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs 2> gpurun_out/r5a_bench_$BATCH.err | tail -1 > gpurun_out/r5a_bench_$BATCH.json
done
tail -n 3 gpurun_out/r5a_match_tests.txt gpurun_out/r5a_mfma_point.txt
cat gpurun_out/r5a_match_prof.jsonl
for BATCH in 4096 8192 16384; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r5a_bench_$BATCH.json').read())
print($BATCH, d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('kernel_ms_per_step'))
"; done
