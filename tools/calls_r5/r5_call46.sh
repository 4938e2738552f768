#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -v "^$" | head -30
python bench.py --no-extra-legs 2>/dev/null | cut -c1-420
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|power" | head
build_tmp/bench_f64_rate
} > gpurun_out/r5_box_check.txt 2>&1
cat gpurun_out/r5_box_check.txt | cut -c1-300
