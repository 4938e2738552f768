#!/bin/bash
# On the GPU box: HBM traffic counters of the pipelined local BA's kernels on a batch (tools/ba_batch_prof.py), one rocprofv3 --pmc
# pass per counter:   bash tools/pmc_ba_batch.sh [P F L B] > gpurun_out/<tag>_ba_pmc.txt
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
CFG="${1:-8} ${2:-4} ${3:-1500} ${4:-64}"
O=gpurun_out/pmc_ba; rm -rf $O; mkdir -p $O
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $CNT -d $O/$CNT -o pmc -- python tools/ba_batch_prof.py $CFG 0 1 > $O/$CNT.out 2> $O/$CNT.err
done
python - "$CFG" <<'P'
import glob, sqlite3, sys, collections
print("pipelined local BA, windows P F L B = %s: rocprofv3 --pmc, mean per launch over the launches that did work (>= half of the largest)" % sys.argv[1])
print("(KB as rocprofv3 reports them; HBM bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 - the gfx950 correction of MI355X_MICROARCH.md that tools/pmc_traffic.py applies; with two lanes a launch covers half of the call's windows)")
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob("gpurun_out/pmc_ba/%s/**/*_results.db" % cnt, recursive=True):
        c = sqlite3.connect(db)
        per = collections.defaultdict(float)
        for name, cn, v, disp in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            if "kp_" in name and cn == cnt:
                per[(name.split("(")[0].split("::")[-1], disp)] += v
        by = collections.defaultdict(list)
        for (n, d), v in per.items():
            by[n].append(v)
        for n, vals in sorted(by.items()):
            vals.sort(reverse=True)
            big = [v for v in vals if v >= 0.5 * vals[0]] if vals[0] > 0 else vals
            m = sum(big) / len(big)
            print("%-12s %-18s launches %4d (of %4d)  mean %14.1f KB as counted" % (cnt, n, len(big), len(vals), m))
P
find $O -name "*.db" -delete; rm -rf $O
