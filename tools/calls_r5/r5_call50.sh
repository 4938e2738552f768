#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
for v in ${VARS:-cg40 cg30 cg20}; do
echo "## variant $v (records per round / candidate table per wave: see tools/calls_r5/r5_call50.sh)"
AB_ONLY=1 GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_$v.so python tools/assoc_pad_ab.py 2>/dev/null
done
echo "## the library"
AB_ONLY=1 python tools/assoc_pad_ab.py 2>/dev/null
} > gpurun_out/${OUT:-r5_assoc_occ}.txt 2>&1
cat gpurun_out/${OUT:-r5_assoc_occ}.txt
