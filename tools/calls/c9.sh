#!/bin/bash
# 8-byte packed cells (assoc_cell8) against the 16-byte cells: tests, time on the bench points (interleaved), PMC traffic
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gmm_assoc.py -x -q 2>&1 | tail -4 | tee gpurun_out/r6_c9_tests.txt
for rep in 1 2 3; do
  for c8 in 1 0; do echo "== assoc_cell8 $c8"; GMMLOC_ASSOC_CELL8=$c8 python tools/assoc_time.py 2>&1 | grep "ms per call"; done
done > gpurun_out/r6_c9_assoc_time.txt 2>&1
cat gpurun_out/r6_c9_assoc_time.txt
