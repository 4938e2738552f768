#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do for fz in -1 0; do echo "== pipe_fuse_asm $fz"; GMMLOC_PIPE_FUSE_ASM=$fz python tools/ba_time.py 2>/dev/null | grep "^P"; done; done | tee gpurun_out/r6_c13_fuse_asm.txt
timeout 1200 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -3 | tee gpurun_out/r6_c13_tests.txt
