#!/bin/bash
# chain v2 (fallback, temporal points, two halves): tests, soak, latency
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_match.py tests/test_host_cabi.py -x -q 2>&1 | tail -15 > gpurun_out/r6_c2_tests.txt
cat gpurun_out/r6_c2_tests.txt
timeout 900 python tools/soak_chain.py 150 2>&1 | tail -12 > gpurun_out/r6_c2_soak_chain.txt
cat gpurun_out/r6_c2_soak_chain.txt
timeout 600 python tools/chain_time.py > gpurun_out/r6_c2_chain_time.json 2> gpurun_out/r6_c2_chain_time.err
tail -3 gpurun_out/r6_c2_chain_time.err; cat gpurun_out/r6_c2_chain_time.json
