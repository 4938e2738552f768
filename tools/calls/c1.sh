#!/bin/bash
# baseline of round 6: where a trial goes (DENSE batch + SPREAD single frame), latency, refine-only rate
export TMPDIR=/tmp
V=$PWD/gmmloc_amd/variants
( echo "== dense 256 frames"; GMMLOC_HIP_LIB=$V/lib_baprof.so python tools/prof_ba.py 256 0
  echo "== spread 1 frame"; GMMLOC_HIP_LIB=$V/lib_baprof.so python tools/prof_ba.py 1 1 ) > gpurun_out/r6_c1_prof.txt 2>&1
python tools/lat1.py > gpurun_out/r6_c1_lat1.txt 2>&1
python tools/refine_only.py 4096 3 > gpurun_out/r6_c1_refine.txt 2>&1
python tools/refine_only.py 16384 3 >> gpurun_out/r6_c1_refine.txt 2>&1
tail -5 gpurun_out/r6_c1_prof.txt gpurun_out/r6_c1_lat1.txt gpurun_out/r6_c1_refine.txt
