set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
V=gmmloc_amd/variants
for i in 1 2; do
for L in gmmloc_amd/libgmmloc_hip.so $V/lib_iilp.so $V/lib_iminreg.so $V/lib_imaxocc.so $V/lib_iilp_pf.so $V/lib_iilp_trk.so $V/lib_iilp_nopost.so $V/lib_iilp_nolsr.so $V/lib_iilp_licm.so; do
  echo "== $L"
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 1 2>/dev/null | tail -1
done; done
echo "== PROF base"; GMMLOC_HIP_LIB=$PWD/$V/lib_prof.so python tools/prof_ba.py 256 2>/dev/null
echo "== PROF iilp"; GMMLOC_HIP_LIB=$PWD/$V/lib_prof_iilp.so python tools/prof_ba.py 256 2>/dev/null
for L in gmmloc_amd/libgmmloc_hip.so $V/lib_iilp_all.so; do
  echo "== $L"
  GMMLOC_HIP_LIB=$PWD/$L python tools/ba_time.py 2>/dev/null | grep "^P"
  GMMLOC_HIP_LIB=$PWD/$L python tools/latency.py 2>/dev/null | tail -12
  GMMLOC_HIP_LIB=$PWD/$L python tools/fixed_time.py 4096 300 2 2>/dev/null | tail -3
done
