#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## k_ba1_fast, SPREAD (one frame, 8 workgroups): phase clocks of workgroup 0 / thread 0"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_baprof.so python tools/prof_ba.py 1 1 2>/dev/null | head -16
echo "## pose tests on the final kernel"
python -m pytest tests/test_gpu_pose.py tests/test_gpu_chain.py -q -x 2>&1 | tail -3
python tools/chain_time.py 2>/dev/null | cut -c1-900
} > gpurun_out/r5_spread_prof.txt 2>&1
cat gpurun_out/r5_spread_prof.txt
