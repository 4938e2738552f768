#!/bin/bash
# A/B of the trial loop's hand-over: v0 barrier / no prefetch, v1 dx word, v2 prefetch, v3 both; default = v3 + blocked solve
export TMPDIR=/tmp
V=$PWD/gmmloc_amd/variants
for rep in 1 2; do
for L in $V/lib_v0.so $V/lib_v1.so $V/lib_v2.so $V/lib_v3.so $PWD/gmmloc_amd/libgmmloc_hip.so; do
  echo "== $(basename $L)"
  GMMLOC_HIP_LIB=$L python tools/refine_only.py 4096 3 2>&1 | grep refine
done
done > gpurun_out/r6_c4_ab.txt 2>&1
for L in $V/lib_v0.so $PWD/gmmloc_amd/libgmmloc_hip.so; do echo "== $(basename $L)"; GMMLOC_HIP_LIB=$L python tools/lat1.py 2>&1 | grep -v amdgpu; done >> gpurun_out/r6_c4_ab.txt
cat gpurun_out/r6_c4_ab.txt
timeout 1500 python -m pytest tests/test_gpu_track.py tests/test_gpu_anchor.py tests/test_gpu_soak_cases.py -q 2>&1 | tail -8 | tee gpurun_out/r6_c4_tests.txt
