#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
python -m pytest tests/test_gpu_gmm_assoc.py tests/test_gpu_track.py -q -x 2>&1 | tail -2
AB_ONLY=1 python tools/assoc_pad_ab.py 2>/dev/null
python tools/run_configs.py --configs 2,5 2>/dev/null | cut -c1-900
} > gpurun_out/r5_assoc_lone.txt 2>&1
cat gpurun_out/r5_assoc_lone.txt
