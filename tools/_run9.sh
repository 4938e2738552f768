set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_match.py -m gpu -q 2>&1 | tail -3
python tools/run_configs.py --configs match 2>/dev/null | tail -4 | cut -c1-400
V=gmmloc_amd/variants
for i in 1 2; do
for L in gmmloc_amd/libgmmloc_hip.so $V/lib_trk.so $V/lib_nolsr.so $V/lib_nosink.so $V/lib_ifcvt.so; do
  echo "== $L"
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 0 2>/dev/null | tail -1
  GMMLOC_HIP_LIB=$PWD/$L python tools/refine_only.py 4096 3 1 2>/dev/null | tail -1
done; done
O=gpurun_out/pmc_schur; rm -rf $O; mkdir -p $O
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $CNT -d $O/$CNT -o pmc -- python tools/ba_batch_prof.py 8 4 1500 64 2 1 > /dev/null 2> $O/$CNT.err
done
python - <<'P'
import glob, sqlite3
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob("gpurun_out/pmc_schur/%s/**/*_results.db" % cn, recursive=True):
        c = sqlite3.connect(db)
        per = {}
        for name, cnt, v, disp in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            k = name.split("(")[0].split("::")[-1]
            per.setdefault((k, disp), 0.0); per[(k, disp)] += v
        by = {}
        for (k, d), v in per.items(): by.setdefault(k, []).append(v)
        for k, vals in sorted(by.items()):
            if k.startswith("kp_"): print(cn, k, "launches", len(vals), "mean per launch (KB as counted)", sum(vals) / len(vals))
P
find $O -name "*.db" -delete
