#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo default; python tools/match_legs.py --legs proj,frame --reps 20 | cut -c1-140
echo desc_lds; GMMLOC_MATCH_DESC_LDS=1 python tools/match_legs.py --legs proj,frame --reps 20 | cut -c1-140
