#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_baprof.so python tools/prof_ba.py 256 0 > gpurun_out/r5g_prof_ba.txt 2>&1
cat gpurun_out/r5g_prof_ba.txt
