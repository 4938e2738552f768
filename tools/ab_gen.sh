#!/bin/bash
# A/B of k_ba_gen builds on one box: tools/ab_gen.sh lib_a.so lib_b.so ...   (paths relative to the repo root)
for i in 1 2; do for L in "$@"; do echo "== $L"; GMMLOC_HIP_LIB=$PWD/$L python tools/ba_time.py 2>&1 | grep "^P"; done; done
