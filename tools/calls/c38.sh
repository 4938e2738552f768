#!/bin/bash
# long soaks on the final library (the evidence set's soaks with more rounds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/soak_match.py 4000 2>/dev/null | tail -3 > gpurun_out/r6j_long_soak_match.txt
python tools/soak_chain.py 8000 2>/dev/null | tail -6 > gpurun_out/r6j_long_soak_chain.txt
python tools/soak.py 6000 > gpurun_out/r6j_long_soak_strict.txt 2>&1
python tools/soak_track.py 500000 > gpurun_out/r6j_long_soak_track.txt 2>&1
tail -1 gpurun_out/r6j_long_soak_match.txt | cut -c1-400; tail -1 gpurun_out/r6j_long_soak_chain.txt | cut -c1-300; tail -1 gpurun_out/r6j_long_soak_strict.txt | cut -c1-500; tail -1 gpurun_out/r6j_long_soak_track.txt | cut -c1-400
