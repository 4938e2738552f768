#!/bin/bash
# per-wave issue / wait shares of k_ba1_fast for one library:  bash tools/pmc_waves.sh <frames> <points> (GMMLOC_HIP_LIB selects the build)
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
NF=${1:-4096}; M=${2:-1984}
O=gpurun_out/pmc_waves; rm -rf $O; mkdir -p $O
i=0
for CNT in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LEVEL_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CNT -d $O/p$i -o pmc -- python tools/refine_only.py $NF 2 0 $M > $O/p$i.out 2> $O/p$i.err
done
python - <<'P'
import glob, sqlite3, collections
out = collections.OrderedDict()
for db in sorted(glob.glob("gpurun_out/pmc_waves/p*/**/*_results.db", recursive=True)):
    c = sqlite3.connect(db)
    try: rows = list(c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"))
    except Exception as e: print("cannot read", db, e); continue
    per = {}
    for name, cn, v, disp in rows:
        if "k_ba1_fast" not in name: continue
        per[(cn, disp)] = per.get((cn, disp), 0.0) + v
    by = {}
    for (cn, disp), v in per.items(): by.setdefault(cn, []).append(v)
    for cn, vals in by.items():
        vals.sort(reverse=True); big = [v for v in vals if v >= 0.5 * vals[0]] if vals[0] > 0 else vals
        out[cn] = sum(big) / len(big)
wc = out.get("SQ_WAVE_CYCLES", 1)
for k, v in out.items(): print("%-26s %16.0f  %.4f of SQ_WAVE_CYCLES" % (k, v, v / wc))
if out.get("SQ_BUSY_CYCLES"): print("mean waves in flight per SQ-busy cycle (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES): %.2f" % (out.get("SQ_LEVEL_WAVES", 0) / out["SQ_BUSY_CYCLES"]))
P
rm -rf $O
