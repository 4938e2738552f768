#!/bin/bash
export TMPDIR=/tmp
python tools/pose_m_sweep.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6_c17_pose_m_sweep.txt
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6_c17_tests.txt
timeout 900 python tools/soak_chain.py 200 2>&1 | tail -6 | tee gpurun_out/r6_c17_soak_chain.txt
timeout 600 python tools/chain_time.py > gpurun_out/r6_c17_chain_time.json 2> gpurun_out/r6_c17_chain_time.err; tail -2 gpurun_out/r6_c17_chain_time.err; cat gpurun_out/r6_c17_chain_time.json
