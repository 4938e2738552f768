#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "# k_ba1_fast, 2 000-point class: today's shape (8 waves, 80 B of LDS per point, one frame per CU) against option ba_two_frames (bafd2000x: 4 waves own two groups"
echo "# each, 32 B of LDS per point, the 48-byte hand-over slot of a point in GLOBAL memory, two frames per CU); same bits"
echo "## time (tools/refine_only.py <frames> 3 0 <points>)"
cat gpurun_out/r5_two_frames.txt
echo "## counters, today's shape (tools/pmc_waves.sh 4096 2000)"
bash tools/pmc_waves.sh 4096 2000
echo "## counters, ba_two_frames = 1"
GMMLOC_BA_TWO_FRAMES=1 bash tools/pmc_waves.sh 4096 2000
} > gpurun_out/r5_two_frames_full.txt 2>&1
tail -45 gpurun_out/r5_two_frames_full.txt
GMMLOC_BA_TWO_FRAMES=0 timeout 600 python -m pytest tests/test_gpu_track.py -x -q -m gpu -k two_frames 2>&1 | tail -3
