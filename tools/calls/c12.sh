#!/bin/bash
export TMPDIR=/tmp
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_pipeprof.so python tools/ba_batch_prof.py 8 4 1500 1 0 1 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/r6_c12_pipeprof.txt
cat gpurun_out/r6_c12_pipeprof.txt
python tools/ba_time.py 2>/dev/null | grep "^P" | head -3
