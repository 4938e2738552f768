#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
python -m pytest tests/test_gpu_gmm_assoc.py -q -x 2>&1 | tail -3
for v in ${VARS:-bal240 bal50}; do
echo "## variant $v"
AB_ONLY=1 GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_$v.so python tools/assoc_pad_ab.py 2>/dev/null
done
echo "## the library"
python tools/assoc_pad_ab.py 2>/dev/null
} > gpurun_out/${OUT:-r5_assoc_bal}.txt 2>&1
cat gpurun_out/${OUT:-r5_assoc_bal}.txt
