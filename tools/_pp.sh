cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BAGEN_MODE=2 python tools/ba_bits.py > gpurun_out/bits_new.txt 2>&1
echo "== new"; BAGEN_MODE=2 python tools/ba_time.py 2>&1 | grep "^P"
timeout 900 python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -3
