#!/bin/bash
export TMPDIR=/tmp
V=$PWD/gmmloc_amd/variants
for rep in 1 2; do
for L in $PWD/gmmloc_amd/libgmmloc_hip.so $V/lib_blk.so; do
  echo "== $(basename $L)"
  GMMLOC_HIP_LIB=$L python tools/lat1.py 2>&1 | grep -v amdgpu
  GMMLOC_HIP_LIB=$L python tools/refine_only.py 4096 3 2>&1 | grep refine
done; done > gpurun_out/r6_c6_blk.txt 2>&1
cat gpurun_out/r6_c6_blk.txt
