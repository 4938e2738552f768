#!/bin/bash
# PC sampling of the refine kernel on the GPU box: where do the waves of k_ba1_fast sit?   tools/pc_sample.sh <tag> [args of refine_only.py]
# (PCS_CMD="python tools/match_legs.py --legs proj --B 256" tools/pc_sample.sh match: another command's dominant kernel)
# -> gpurun_out/<tag>_pc_hist.txt (samples per instruction of the dominant kernel, tools/pc_hist.py)
set -u
TAG=${1:-pc}
shift
OUT=gpurun_out/pcs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for METHOD in stochastic host_trap; do
  UNIT=cycles; INT=1048576
  if [ $METHOD = host_trap ]; then UNIT=time; INT=1; fi
  rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INT --kernel-trace \
    --output-format csv -d $OUT/$METHOD -o pcs -- ${PCS_CMD:-python tools/refine_only.py} "$@" > $OUT/$METHOD.log 2>&1
  echo "== $METHOD rc=$?"; tail -3 $OUT/$METHOD.log
  ls -la $OUT/$METHOD/* 2>/dev/null | head
  F=$(ls $OUT/$METHOD/*/*pc_sampling*.csv $OUT/$METHOD/*pc_sampling*.csv 2>/dev/null | head -1)
  if [ -n "$F" ]; then
    head -3 "$F"
    python tools/pc_hist.py "$F" > gpurun_out/${TAG}_pc_hist_$METHOD.txt 2>&1
    head -60 gpurun_out/${TAG}_pc_hist_$METHOD.txt
    break
  fi
done
rm -rf $OUT/*/*.db
