#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gmm_assoc.py -x -q 2>&1 | tail -3 | tee gpurun_out/r6_c11_tests.txt
for rep in 1 2 3; do for c8 in 1 0; do
  GMMLOC_ASSOC_CELL8=$c8 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('assoc_cell8 $c8', 'value', round(d['value']), d['kernel_ms_per_step'], d.get('assoc_index',{}).get('packed_cell_bytes'))"
done; done | tee gpurun_out/r6_c11_cell8_bench.txt
