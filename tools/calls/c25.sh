#!/bin/bash
# PC sampling of k_search_by_projection<0, true> (one frame per CU, 3 000 map points)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_matchg.so MATCH_LEGS_NP=3000
PCS_CMD="python tools/match_legs.py --legs proj --B 256 --reps 200" bash tools/pc_sample.sh match 2>&1 | tail -80
