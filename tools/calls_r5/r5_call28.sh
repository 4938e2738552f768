#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python tools/soak_chain.py 150 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r5_soak_chain.txt
cat gpurun_out/r5_soak_chain.txt
