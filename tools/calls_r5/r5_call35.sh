#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "# long soaks on the final library of round 5 (same scripts as tools/final_round.sh, more rounds)"
python tools/soak_match.py 5000 2>/dev/null | tail -3
python tools/soak_chain.py 8000 2>/dev/null | tail -6
python tools/soak.py 8000 2>&1 | grep -v amdgpu.ids | tail -12
python tools/soak_track.py 300000 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r5h_long_soak.txt 2>&1
tail -8 gpurun_out/r5h_long_soak.txt | cut -c1-600
