#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## BA tests (Newton reciprocal / rsqrt per observation)"
python -m pytest tests/test_gpu_ba.py tests/test_gpu_soak_cases.py -q -x 2>&1 | tail -4
echo "## ba_time before"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_babefore.so python tools/ba_time.py 2>/dev/null | grep "^P"
echo "## ba_time after"
python tools/ba_time.py 2>/dev/null | grep "^P"
echo "## ba_time before (again)"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_babefore.so python tools/ba_time.py 2>/dev/null | grep "^P"
echo "## ba_time after (again)"
python tools/ba_time.py 2>/dev/null | grep "^P"
} > gpurun_out/r5_ba_nr.txt 2>&1
cat gpurun_out/r5_ba_nr.txt
