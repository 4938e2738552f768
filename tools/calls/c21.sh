#!/bin/bash
# fused chain launches: tests, timeline, latency
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_pose.py tests/test_gpu_adapter.py -x -q -m gpu > gpurun_out/r6_c21_tests.txt 2>&1
tail -5 gpurun_out/r6_c21_tests.txt
rm -rf gpurun_out/chaintrace gpurun_out/chaintrace_fb
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace -- python tools/chain_trace.py run > gpurun_out/r6_c21_run.txt 2>&1
python tools/chain_trace.py table gpurun_out/chaintrace > gpurun_out/r6_chain_trace_fused.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace_fb -- python tools/chain_trace.py run fallback >> gpurun_out/r6_c21_run.txt 2>&1
python tools/chain_trace.py table gpurun_out/chaintrace_fb > gpurun_out/r6_chain_trace_fused_fb.txt 2>&1
cat gpurun_out/r6_chain_trace_fused.txt
timeout 600 python tools/chain_time.py > gpurun_out/r6_c21_chain_time.txt 2>&1
tail -30 gpurun_out/r6_c21_chain_time.txt
timeout 300 python tools/pose_m_sweep.py > gpurun_out/r6_c21_pose_m_sweep.txt 2>&1
cat gpurun_out/r6_c21_pose_m_sweep.txt
