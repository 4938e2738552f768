#!/bin/bash
export TMPDIR=/tmp
python tools/refine_only.py 4096 3 > gpurun_out/r6_c3_refine.txt 2>&1
python tools/refine_only.py 16384 3 >> gpurun_out/r6_c3_refine.txt 2>&1
python tools/refine_only.py 4096 3 1 >> gpurun_out/r6_c3_refine.txt 2>&1
python tools/lat1.py >> gpurun_out/r6_c3_refine.txt 2>&1
cat gpurun_out/r6_c3_refine.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_gpu_track.py tests/test_gpu_anchor.py tests/test_gpu_soak_cases.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6_c3_tests.txt
