#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_soak_cases.py tests/test_gpu_anchor.py tests/test_gpu_threads.py -x -q 2>&1 | tail -3 | tee gpurun_out/r6_c16_tests.txt
BAGEN_MODE=1 python tools/ba_time.py 2>/dev/null | grep "^P" | head -3
python tools/ba_time.py 2>/dev/null | grep "^P"
