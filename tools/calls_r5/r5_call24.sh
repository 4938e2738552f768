#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
A=$PWD/gmmloc_amd/variants/lib_genbefore.so
{
echo "# k_ba_gen with the problem's addresses in an LDS copy of GenP, taken again after every problem-wide barrier (36 -> 0 spilled VGPRs, 1 310 -> 1 024 spilled SGPRs, scratch 96 -> 24 B)"
echo "## bits (tools/ba_bits.py: 54 windows, persistent kernel forced / default route): before | after"
GMMLOC_HIP_LIB=$A BAGEN_MODE=1 python tools/ba_bits.py 2>&1 | tail -2
BAGEN_MODE=1 python tools/ba_bits.py 2>&1 | tail -2
GMMLOC_HIP_LIB=$A python tools/ba_bits.py 2>&1 | tail -1
python tools/ba_bits.py 2>&1 | tail -1
echo "## time, persistent kernel forced (BAGEN_MODE=1 tools/ba_time.py), before / after, twice"
for i in 1 2; do
echo before; GMMLOC_HIP_LIB=$A BAGEN_MODE=1 python tools/ba_time.py 2>&1 | grep "^P"
echo after; BAGEN_MODE=1 python tools/ba_time.py 2>&1 | grep "^P"
done
echo "## windows below the pipelined shape's threshold (tools/ba_modes.py), before / after"
echo before; GMMLOC_HIP_LIB=$A python tools/ba_modes.py 2>&1 | tail -12
echo after; python tools/ba_modes.py 2>&1 | tail -12
} > gpurun_out/r5_ba_gen_fresh.txt 2>&1
cat gpurun_out/r5_ba_gen_fresh.txt
timeout 900 python -m pytest tests/test_gpu_ba.py -x -q -m gpu 2>&1 | tail -3
