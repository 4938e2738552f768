#!/bin/bash
# tools/build_variant.sh NAME "FLAGS" file.hip [file2.hip ...]: gmmloc_amd/variants/lib_NAME.so = the in-tree library with the named
# sources rebuilt with extra FLAGS (A/B and profiling builds; run with GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_NAME.so)
set -e
NAME=$1; FL=$2; shift 2
cd "$(dirname "$0")/../gmmloc_amd/csrc"
make -s all
mkdir -p ../variants /tmp/glv_$NAME
OBJS=""
for o in *.o; do
  src=${o%.o}.hip
  hit=0; for f in "$@"; do [ "$f" = "$src" ] && hit=1; done
  if [ $hit = 1 ]; then
    # the per-file LLVM flags of the Makefile (NOLICM list, gl_ba_fast.o): a variant differs from the library by $FL alone
    PF=""
    case " gl_ba_fast.hip gl_ba_gen.hip gl_ba.hip gl_refine_pose.hip gl_point.hip gl_view.hip gl_match.hip " in *" $src "*) PF="-mllvm -disable-machine-licm";; esac
    [ "$src" = "gl_ba_fast.hip" ] && PF="$PF -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -disable-machine-sink"
    /opt/rocm/bin/hipcc $FL $PF --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. -c $src -o /tmp/glv_$NAME/$o
    OBJS="$OBJS /tmp/glv_$NAME/$o"
  else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$NAME.so $OBJS
echo built gmmloc_amd/variants/lib_$NAME.so
