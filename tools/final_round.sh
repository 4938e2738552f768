#!/bin/bash
# Everything the round's numbers come from, on one box: tools/final_round.sh <tag>   -> gpurun_out/<tag>_*
set -u
TAG=${1:-r6}
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.txt
# (tools/kernel_meta.py reads the object files of the build container: run it there -> profiles/<tag>_kernel_meta.txt)
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
rm -rf gpurun_out/prof_$TAG   # (the rocprofv3 databases: summarised above; gpurun copies at most 64 MiB back)
cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json   # (on the box: bench.py's `roofline.traffic` reads the latest profiles/r*_traffic.json - this run's own, same sources)
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
python tools/run_configs.py --configs 2,3,4,5,ba,kf,match > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err
python tools/latency.py > gpurun_out/${TAG}_latency.txt 2>/dev/null
python tools/replay_euroc.py --anchor prior > gpurun_out/${TAG}_replay_euroc.json 2>/dev/null
ANCHORS="none prior" SIGMAS="0" bash tools/replay_matrix.sh > gpurun_out/${TAG}_replay_matrix.txt 2>/dev/null
ANCHORS="fixed" SIGMAS="0 0.02" bash tools/replay_matrix.sh >> gpurun_out/${TAG}_replay_matrix.txt 2>/dev/null
python tools/ba_time.py 2>/dev/null | grep "^P" > gpurun_out/${TAG}_ba_time.txt
python tools/ba_modes.py 2>/dev/null | grep "^P" > gpurun_out/${TAG}_ba_modes.txt
python tools/fixed_time.py 4096 300 2 > gpurun_out/${TAG}_fixed_time.txt 2>/dev/null
python tools/fixed_time.py 2048 1000 2 >> gpurun_out/${TAG}_fixed_time.txt 2>/dev/null
python tools/fixed_time.py 1024 2000 2 >> gpurun_out/${TAG}_fixed_time.txt 2>/dev/null
python tools/lat1.py 2>/dev/null | grep -v amdgpu > gpurun_out/${TAG}_lat1.txt
python tools/chain_time.py > gpurun_out/${TAG}_chain_time.json 2>/dev/null
python tools/chain_ab.py $TAG 2>/dev/null | grep label > gpurun_out/${TAG}_chain_device_time.json
bash tools/pmc_match.sh ${TAG} > gpurun_out/${TAG}_pmc_match.log 2>&1
rm -rf gpurun_out/pmc_match_${TAG}
python tools/soak_match.py ${SOAK_MATCH:-300} 2>/dev/null | tail -3 > gpurun_out/${TAG}_soak_match.txt
python tools/soak_chain.py ${SOAK_CHAIN:-1500} 2>/dev/null | tail -12 > gpurun_out/${TAG}_soak_chain.txt; tail -1 gpurun_out/${TAG}_soak_chain.txt | cut -c1-400
python tools/soak.py 2000 > gpurun_out/${TAG}_soak_strict.txt 2>&1
tail -2 gpurun_out/${TAG}_soak_strict.txt | cut -c1-500
python tools/soak_track.py ${SOAK_TRACK:-86000} > gpurun_out/${TAG}_soak_track.txt 2>&1
tail -1 gpurun_out/${TAG}_soak_track.txt | cut -c1-500
cat gpurun_out/${TAG}_gpu_tests.txt
