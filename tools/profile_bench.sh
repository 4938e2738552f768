#!/bin/bash
# On the GPU box: rocprofv3 evidence for the bench command (profiles/<tag>_*).
#   tools/profile_bench.sh r2          -> gpurun_out/prof_<tag>/..., summaries copied to gpurun_out/<tag>_*.txt|json
# Pass 1: --kernel-trace --stats of the SAME command bench.py is run with (fewer steps);
# passes 2..4: --pmc counters, each in its own run with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.err
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/${TAG}_bench_kernel_stats.txt
tail -1 $OUT/bench_line_under_rocprof.json > gpurun_out/${TAG}_bench_line_under_rocprof.json
PMC="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc$i -o pmc -- $PMC > /dev/null 2> $OUT/pmc$i.err
done
python tools/pmc_traffic.py $OUT > gpurun_out/${TAG}_traffic.json
cat gpurun_out/${TAG}_traffic.json | head -60
