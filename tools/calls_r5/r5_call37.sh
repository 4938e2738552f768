#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r5i_gpu_tests.txt
python bench.py > gpurun_out/r5i_bench_line.json 2> gpurun_out/r5i_bench.err
cat gpurun_out/r5i_gpu_tests.txt; cut -c1-900 gpurun_out/r5i_bench_line.json
