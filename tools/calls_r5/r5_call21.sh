#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/pmc_match.sh r5f > gpurun_out/r5f_pmc_match.log 2>&1
rm -rf gpurun_out/pmc_match_r5f
cut -c1-200 gpurun_out/r5f_match_legs.jsonl
