#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_baprofl.so python tools/prof_ba.py 4096 0 2>&1 | grep -v amdgpu.ids | tail -16 | tee gpurun_out/r5m_prof_stations.txt
python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1
