#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "# long soaks on the FINAL library of round 5 (same scripts as tools/final_round.sh, more rounds)"
python tools/soak_match.py 4000 2>/dev/null | tail -3
python tools/soak_chain.py 6000 2>/dev/null | tail -4
python tools/soak.py 6000 2>&1 | grep -v amdgpu.ids | tail -12
python tools/soak_track.py 200000 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r5n_long_soak.txt 2>&1
tail -8 gpurun_out/r5n_long_soak.txt | cut -c1-600
