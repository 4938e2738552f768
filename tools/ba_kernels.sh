#!/bin/bash
# rocprofv3 kernel stats of one local-BA window per size in the pipelined shape (three calls each): tools/ba_kernels.sh > profiles/<tag>_bagen_kernels.txt
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "rocprofv3 --kernel-trace --stats -- python tools/ba_one.py P F L 2   (3 calls of gl_joint_optimization, bagen_mode 2; columns: launches, total us, average us, % of the run's kernel time; then the launch shapes: grid x block, VGPRs, LDS, scratch)"
for cfg in "8 4 1500" "12 4 2000" "20 8 3000"; do
  rm -rf gpurun_out/pp; rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o t -- python tools/ba_one.py $cfg 2 > /dev/null 2>&1
  DB=$(ls gpurun_out/pp/*/*_results.db gpurun_out/pp/*_results.db 2>/dev/null | head -1)
  echo; echo "== free + fixed key-frames / points: $cfg"; python tools/rocpd_summary.py "$DB" | grep "kp_" 
done
