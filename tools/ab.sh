#!/bin/bash
# A/B two builds of libgmmloc_hip.so on the same GPU box (interleaved): tools/ab.sh libA.so libB.so [bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    GMMLOC_HIP_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), d['kernel_ms_per_step'])"
  done
done
