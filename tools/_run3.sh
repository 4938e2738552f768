set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
V=gmmloc_amd/variants
for i in 1 2; do
for L in gmmloc_amd/libgmmloc_hip.so $V/lib_pf.so $V/lib_pf_ilp.so $V/lib_ilp.so $V/lib_iilp.so; do
  echo "== $L"
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 1 2>/dev/null | tail -1
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 1000 2>/dev/null | tail -1
done; done
