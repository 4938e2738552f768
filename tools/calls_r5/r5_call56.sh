#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
python -m pytest tests/test_gpu_gmm_assoc.py tests/test_gpu_track.py -q -x 2>&1 | tail -2
echo "## the library (four tiles per workgroup at 4 waves per SIMD: 128 registers, 10 spilled)"
AB_ONLY=1 python tools/assoc_pad_ab.py 2>/dev/null
echo "## variant wpe3 (four tiles per workgroup at 3 waves per SIMD: 144 registers, none spilled)"
AB_ONLY=1 GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_wpe3.so python tools/assoc_pad_ab.py 2>/dev/null
python bench.py --no-extra-legs --no-cpu-baseline 2>/dev/null | cut -c1-330
} > gpurun_out/r5_assoc_tiles.txt 2>&1
cat gpurun_out/r5_assoc_tiles.txt
