#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "# searchByProjection in batches: descriptors in LDS (1 024 threads per frame) against the batch shape; station clocks"
echo "## default"
python tools/match_legs.py --legs proj,frame --reps 10 2>/dev/null
echo "## GMMLOC_MATCH_DESC_LDS=1"
GMMLOC_MATCH_DESC_LDS=1 python tools/match_legs.py --legs proj,frame --reps 10 2>/dev/null
echo "## stations (GL_MATCH_PROF build), default shape"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so python tools/match_legs.py --legs proj,frame --reps 5 --prof 2>/dev/null
echo "## stations, descriptors in LDS"
GMMLOC_MATCH_DESC_LDS=1 GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchprof.so python tools/match_legs.py --legs proj,frame --reps 5 --prof 2>/dev/null
echo "## B=256 (one frame per CU) default / desc lds"
python tools/match_legs.py --legs proj,frame --reps 10 --B 256 2>/dev/null
GMMLOC_MATCH_DESC_LDS=0 python tools/match_legs.py --legs proj,frame --reps 10 --B 256 2>/dev/null
} > gpurun_out/r5_match_dl.txt 2>&1
cat gpurun_out/r5_match_dl.txt | cut -c1-700
