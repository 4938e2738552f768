cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BAGEN_MODE=2 python tools/ba_bits.py > gpurun_out/bits_new.txt 2>&1
echo "== new"; BAGEN_MODE=2 python tools/ba_time.py 2>&1 | grep "^P"
timeout 900 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -3
for cfg in "20 8 3000" "8 4 1500"; do
  rm -rf gpurun_out/pp; rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o t -- python tools/ba_one.py $cfg 2 > /dev/null 2>&1
  DB=$(ls gpurun_out/pp/*/*_results.db gpurun_out/pp/*_results.db 2>/dev/null | head -1)
  echo "== kernels $cfg"; python tools/rocpd_summary.py "$DB" | grep "kp_\|k_ba" | head -5
done
