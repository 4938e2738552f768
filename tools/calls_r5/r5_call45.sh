#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
SOAK_MATCH=1000 SOAK_CHAIN=4000 bash tools/final_round.sh r5m > gpurun_out/r5m_final_round.log 2>&1
tail -5 gpurun_out/r5m_final_round.log | cut -c1-600
cat gpurun_out/r5m_bench_line.json | cut -c1-600
