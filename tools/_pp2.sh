cd $GRAFT_REPO_ROOT
for cfg in "8 4 1500" "20 8 3000"; do
  echo "== $cfg"; GMMLOC_HIP_LIB=$PWD/gmmloc_amd/variants/lib_pipeprof.so python tools/ba_one.py $cfg 2 2>&1 | grep "cycles:" | tail -1
done
