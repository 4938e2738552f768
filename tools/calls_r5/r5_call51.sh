#!/bin/bash
# the headline evidence again on the FINAL library (the association's gather changed after r5m): tests, rocprof + PMC passes, bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
TAG=r5n
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.txt
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
rm -rf gpurun_out/prof_$TAG
cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
python tools/run_configs.py --configs 2,5 > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err
python tools/soak.py 500 2>&1 | tail -2 | cut -c1-500 > gpurun_out/${TAG}_soak_strict.txt
cat gpurun_out/${TAG}_gpu_tests.txt; cut -c1-400 gpurun_out/${TAG}_bench_line.json
