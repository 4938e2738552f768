#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pose.py tests/test_gpu_chain.py tests/test_gpu_adapter.py tests/test_gpu_replay.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6_c18_tests.txt
for m in -1 0; do echo "== pose_compact $m"; GMMLOC_POSE_COMPACT=$m python tools/pose_m_sweep.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r6_c18_pose_m_sweep.txt
timeout 600 python tools/chain_time.py > gpurun_out/r6_c18_chain_time.json 2> gpurun_out/r6_c18_chain_time.err; tail -2 gpurun_out/r6_c18_chain_time.err; python -c "
import json; d=json.load(open('gpurun_out/r6_c18_chain_time.json')); print({k:d[k] for k in d if 'ms' in k or 'frames_per_s' in k})"
