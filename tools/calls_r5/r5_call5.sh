#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
./build_tmp/bench_mfma_point 2000 > gpurun_out/r5e_mfma_point.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r5e_chain_tests.txt
timeout 600 python tools/chain_time.py > gpurun_out/r5e_chain_time.json 2> gpurun_out/r5e_chain_time.err
cat gpurun_out/r5e_mfma_point.txt gpurun_out/r5e_chain_tests.txt gpurun_out/r5e_chain_time.json; tail -5 gpurun_out/r5e_chain_time.err
