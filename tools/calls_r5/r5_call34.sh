#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/refine_only.py 4096 3 0 2000 2>&1 | grep "^refine"
GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 4096 3 0 2000 2>&1 | grep "^refine"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_unnt.so GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 4096 3 0 2000 2>&1 | grep "^refine"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_unnt.so GMMLOC_BA_TWO_FRAMES=1 python tools/refine_only.py 16384 3 0 2000 2>&1 | grep "^refine"
