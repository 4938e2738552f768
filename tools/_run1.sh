set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/s2_gpu_tests.txt
for m in 0 1 3; do python tools/ba_batch_prof.py 8 4 1500 64 $m 2; done > gpurun_out/s2_ba_batch.txt 2>&1
python tools/ba_batch_prof.py 8 4 1500 256 0 1 >> gpurun_out/s2_ba_batch.txt 2>&1
python tools/ba_batch_prof.py 8 4 1500 256 3 1 >> gpurun_out/s2_ba_batch.txt 2>&1
for m in 2 3; do
rm -rf gpurun_out/pp; rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o t -- python tools/ba_batch_prof.py 8 4 1500 64 $m 2 > /dev/null 2>&1
DB=$(ls gpurun_out/pp/*/*_results.db gpurun_out/pp/*_results.db 2>/dev/null | head -1)
echo "== mode $m B 64"; python tools/rocpd_summary.py "$DB" | grep "kp_\|k_ba_gen"
done > gpurun_out/s2_ba_batch_kernels.txt 2>&1
rm -rf gpurun_out/pp
python bench.py --no-cpu-baseline > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
cat gpurun_out/s2_gpu_tests.txt gpurun_out/s2_ba_batch.txt gpurun_out/s2_ba_batch_kernels.txt; tail -c 600 gpurun_out/s2_bench.json
