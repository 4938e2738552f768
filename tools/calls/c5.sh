#!/bin/bash
export TMPDIR=/tmp
V=$PWD/gmmloc_amd/variants
( echo "== spread 1 frame (prof build)"; GMMLOC_HIP_LIB=$V/lib_baprof.so python tools/prof_ba.py 1 1 2>&1 | head -9 ) > gpurun_out/r6_c5_prof_spread.txt 2>&1
cat gpurun_out/r6_c5_prof_spread.txt
python tools/latency.py 2>/dev/null | head -12 | tee gpurun_out/r6_c5_latency.txt
