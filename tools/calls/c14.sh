#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do for L in $PWD/gmmloc_amd/libgmmloc_hip.so $PWD/gmmloc_amd/variants/lib_slowstep.so; do echo "== $(basename $L)"; GMMLOC_HIP_LIB=$L python tools/ba_time.py 2>/dev/null | grep "^P" | head -3; done; done | tee gpurun_out/r6_c14_fast_step.txt
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_soak_cases.py -x -q 2>&1 | tail -3 | tee gpurun_out/r6_c14_tests.txt
