#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/assoc_cell_sweep.py 0.03 0.04 0.05 0.07 0.1 0.15 0.2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5q_assoc_cell_sweep.txt
