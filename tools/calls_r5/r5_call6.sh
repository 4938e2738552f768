#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r5f_gpu_tests.txt
cat gpurun_out/r5f_gpu_tests.txt
