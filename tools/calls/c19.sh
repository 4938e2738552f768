#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do for L in $PWD/gmmloc_amd/libgmmloc_hip.so $PWD/gmmloc_amd/variants/lib_prepold.so; do echo "== $(basename $L)"; GMMLOC_HIP_LIB=$L python tools/track_sparse.py 2>&1 | grep -v amdgpu; done; done | tee gpurun_out/r6_c19_track_sparse.txt
timeout 2000 python -m pytest tests/test_gpu_track.py tests/test_gpu_anchor.py tests/test_gpu_soak_cases.py tests/test_gpu_chain.py tests/test_gpu_replay.py tests/test_gpu_adapter.py tests/test_gpu_threads.py -x -q 2>&1 | tail -4 | tee gpurun_out/r6_c19_tests.txt
timeout 600 python tools/soak_track.py 20000 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/r6_c19_soak_track.txt
timeout 600 python tools/chain_time.py > gpurun_out/r6_c19_chain_time.json 2> gpurun_out/r6_c19_chain_time.err; tail -2 gpurun_out/r6_c19_chain_time.err
