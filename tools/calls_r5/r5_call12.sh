#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_nodynb.so python tools/refine_only.py 4096 2500 0 > gpurun_out/r5l_refine.txt 2>&1 &
PID=$!
for i in $(seq 1 45); do
  echo -n "t=$i "; rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Package Power|sclk|GPU use" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done | tee gpurun_out/r5l_smi_busy.txt
wait $PID
tail -1 gpurun_out/r5l_refine.txt
