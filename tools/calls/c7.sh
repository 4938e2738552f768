#!/bin/bash
# which refine build for the latency shape AND the batch: r5 (in-tree), r5 + blocked 6x6 solve (blk), the head/tail + contraction-off structure (x0), x0 + blocked (x0b)
export TMPDIR=/tmp
V=$PWD/gmmloc_amd/variants
for rep in 1 2 3; do
for L in $PWD/gmmloc_amd/libgmmloc_hip.so $V/lib_blk.so $V/lib_x0.so $V/lib_x0b.so; do
  echo "== $(basename $L)"
  GMMLOC_HIP_LIB=$L python tools/lat1.py 2>&1 | grep -v amdgpu
  GMMLOC_HIP_LIB=$L python tools/refine_only.py 16384 3 2>&1 | grep refine
done; done > gpurun_out/r6_c7_pick.txt 2>&1
cat gpurun_out/r6_c7_pick.txt
