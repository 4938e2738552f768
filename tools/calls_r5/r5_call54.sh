#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
for v in ${VARS:-lm1024 lm700}; do
echo "## variant $v"
AB_ONLY=1 GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_$v.so python tools/assoc_pad_ab.py 2>/dev/null | grep "round"
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_$v.so python tools/run_configs.py --configs 5 2>/dev/null | grep -o '"index_ms": [0-9.]*'
done
} > gpurun_out/r5_assoc_lone2.txt 2>&1
cat gpurun_out/r5_assoc_lone2.txt
