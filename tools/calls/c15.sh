#!/bin/bash
export TMPDIR=/tmp
# optimizeCurrentPose with the 6x6 solve by blocks (in-tree) against the sequential one (posepk): latency shapes, batch, chain
for rep in 1 2; do for L in $PWD/gmmloc_amd/libgmmloc_hip.so $PWD/gmmloc_amd/variants/lib_posepk.so; do echo "== $(basename $L)"
  GMMLOC_HIP_LIB=$L python tools/latency.py 2>/dev/null | grep "optimizeCurrentPose"
  GMMLOC_HIP_LIB=$L python tools/chain_time.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('chain one frame', round(d['chain_one_frame_ms'],4), 'mean16', round(d['chain_one_frame_ms_mean_of_16_frames'],4), 'batch frames/s', round(d['chain_batch_frames_per_s']))"
done; done | tee gpurun_out/r6_c15_pose_blocked.txt
timeout 1500 python -m pytest tests/test_gpu_pose.py tests/test_gpu_chain.py tests/test_gpu_ba.py tests/test_gpu_soak_cases.py -x -q 2>&1 | tail -3 | tee gpurun_out/r6_c15_tests.txt
