#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
W3=$PWD/gmmloc_amd/variants/lib_w3.so
{
echo "# k_ba1_fast, 2 000-point class on frames of 1 984 points (31 chunks): today's build (8 waves = 2 per SIMD, groups of 4 chunks, 252 VGPRs, 0 spilled)"
echo "# against -DGL_BAF_W3=1 (11 of 12 waves = 3 per SIMD, groups of 3 chunks, __launch_bounds__(768, 3): 168 VGPRs, 152 spilled, 340 B of scratch per thread)"
echo "## time (tools/refine_only.py 4096 3 0 1984)"
REFINE_SAVE=/tmp/p_def.npy python tools/refine_only.py 4096 3 0 1984
GMMLOC_HIP_LIB=$W3 REFINE_COMPARE=/tmp/p_def.npy python tools/refine_only.py 4096 3 0 1984
REFINE_SAVE=/tmp/p_def.npy python tools/refine_only.py 4096 3 0 1984
GMMLOC_HIP_LIB=$W3 REFINE_COMPARE=/tmp/p_def.npy python tools/refine_only.py 4096 3 0 1984
echo "## counters, today's build (one rocprofv3 --pmc pass per group; 4 096 frames per launch)"
bash tools/pmc_waves.sh 4096 1984
echo "## counters, GL_BAF_W3"
GMMLOC_HIP_LIB=$W3 bash tools/pmc_waves.sh 4096 1984
} > gpurun_out/r5_w3.txt 2>&1
cat gpurun_out/r5_w3.txt
