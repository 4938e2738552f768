#!/bin/bash
# the round's final evidence on the final library (HEAD after the optimizeCurrentPose sessions): profiles/r5m_*
cd ${GRAFT_REPO_ROOT:-/root/repo}
SOAK_MATCH=500 SOAK_CHAIN=1500 SOAK_TRACK=40000 bash tools/final_round.sh r5m > gpurun_out/r5m_final_round.log 2>&1
tail -5 gpurun_out/r5m_final_round.log | cut -c1-600
cat gpurun_out/r5m_bench_line.json | cut -c1-600
