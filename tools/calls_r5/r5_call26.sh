#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so python tools/match_legs.py --legs proj,frame --reps 5 --prof 2>&1 | grep -v amdgpu.ids
