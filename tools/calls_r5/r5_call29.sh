#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p build_tmp
hipcc --offload-arch=gfx950 -O3 -o build_tmp/bench_mfma_point tools/bench_mfma_point.hip 2>/dev/null
SOAK_MATCH=1000 SOAK_CHAIN=1500 bash tools/final_round.sh r5h 2>&1 | tail -30
