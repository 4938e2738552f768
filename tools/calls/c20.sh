#!/bin/bash
# kernel timeline of one chain call (plain, and through the key-frame fallback)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/chaintrace gpurun_out/chaintrace_fb
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace -- python tools/chain_trace.py run > gpurun_out/r6_c20_run.txt 2>&1
python tools/chain_trace.py table gpurun_out/chaintrace > gpurun_out/r6_chain_trace.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace_fb -- python tools/chain_trace.py run fallback >> gpurun_out/r6_c20_run.txt 2>&1
python tools/chain_trace.py table gpurun_out/chaintrace_fb > gpurun_out/r6_chain_trace_fb.txt 2>&1
tail -5 gpurun_out/r6_c20_run.txt
cat gpurun_out/r6_chain_trace.txt
