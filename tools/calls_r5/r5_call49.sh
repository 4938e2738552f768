#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
python tools/assoc_pad_ab.py 2>/dev/null
python -m pytest tests/test_gpu_gmm_assoc.py tests/test_gpu_track.py -q -x 2>&1 | tail -3
python bench.py --no-extra-legs --no-cpu-baseline 2>/dev/null | cut -c1-500
} > gpurun_out/r5_assoc_pad.txt 2>&1
cat gpurun_out/r5_assoc_pad.txt
