set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
V=gmmloc_amd/variants
for i in 1 2; do
for L in gmmloc_amd/libgmmloc_hip.so $V/lib_nolicm_baf.so $V/lib_nolicm_all.so; do
  echo "== $L"
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 1 2>/dev/null | tail -1
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 1000 2>/dev/null | tail -1
done; done
for L in gmmloc_amd/libgmmloc_hip.so $V/lib_nolicm_all.so; do
  echo "== $L"
  GMMLOC_HIP_LIB=$PWD/$L python tools/ba_time.py 2>/dev/null | grep "^P"
  for m in 1 3; do GMMLOC_HIP_LIB=$PWD/$L python tools/ba_batch_prof.py 8 4 1500 64 $m 2 2>/dev/null | tail -1; done
  GMMLOC_HIP_LIB=$PWD/$L python tools/latency.py 2>/dev/null | tail -12
  GMMLOC_HIP_LIB=$PWD/$L python tools/fixed_time.py 4096 300 2 2>/dev/null | tail -3
done
GMMLOC_HIP_LIB=$PWD/$V/lib_nolicm_all.so python -m pytest tests -m gpu -x -q 2>&1 | tail -3
