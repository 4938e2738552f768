#!/bin/bash
# the whole GPU suite on the library with the fused chain launches and the new matcher walk; the chain's timeline; soaks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6_c29_tests.txt 2>&1
tail -3 gpurun_out/r6_c29_tests.txt
rm -rf gpurun_out/chaintrace2 gpurun_out/chaintrace2_fb
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace2 -- python tools/chain_trace.py run > gpurun_out/r6_c29_run.txt 2>&1
python tools/chain_trace.py table gpurun_out/chaintrace2 > gpurun_out/r6_chain_trace_after.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace2_fb -- python tools/chain_trace.py run fallback >> gpurun_out/r6_c29_run.txt 2>&1
python tools/chain_trace.py table gpurun_out/chaintrace2_fb > gpurun_out/r6_chain_trace_after_fb.txt 2>&1
cat gpurun_out/r6_chain_trace_after.txt gpurun_out/r6_chain_trace_after_fb.txt
timeout 900 python tools/soak_chain.py 300 > gpurun_out/r6_c29_soak_chain.txt 2>&1; tail -3 gpurun_out/r6_c29_soak_chain.txt
timeout 600 python tools/chain_time.py > gpurun_out/r6_c29_chain_time.txt 2>&1; tail -2 gpurun_out/r6_c29_chain_time.txt
