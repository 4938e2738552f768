#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for L in gmmloc_amd/variants/lib_nodynb.so gmmloc_amd/libgmmloc_hip.so gmmloc_amd/variants/lib_dynbp0.so; do
    GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1 | sed "s|^|$L plain |"
  done
done 2>&1 | tee gpurun_out/r5k_ab_dynb.txt
