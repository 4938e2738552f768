#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r6_c8_gpu_tests.txt
timeout 900 python tools/soak_track.py 40000 > gpurun_out/r6_c8_soak_track.txt 2>&1; tail -2 gpurun_out/r6_c8_soak_track.txt | cut -c1-600
