#!/bin/bash
# On the GPU box: issue / stall counters of k_ba1_fast on the bench frames (tools/refine_only.py), one rocprofv3 --pmc pass per
# group of four counters:   bash tools/pmc_stalls.sh [frames] [prior] > gpurun_out/<tag>_stalls.txt
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
NF=${1:-1024}; PRIOR=${2:-0}
O=gpurun_out/pmc_stalls; rm -rf $O; mkdir -p $O
W="python tools/refine_only.py $NF 2 $PRIOR"
i=0
for CNT in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CNT -d $O/p$i -o pmc -- $W > $O/p$i.out 2> $O/p$i.err
done
python - <<'P'
import glob, sqlite3, collections
out = collections.OrderedDict()
for db in sorted(glob.glob("gpurun_out/pmc_stalls/p*/**/*_results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"))
    except Exception as e:
        print("cannot read", db, e); continue
    per = {}
    for name, cn, v, disp in rows:
        if "k_ba1_fast" not in name: continue
        per.setdefault((cn, disp), 0.0); per[(cn, disp)] += v
    by = {}
    for (cn, disp), v in per.items(): by.setdefault(cn, []).append(v)
    for cn, vals in by.items():
        vals.sort(reverse=True); big = [v for v in vals if v >= 0.5 * vals[0]] if vals[0] > 0 else vals
        out[cn] = sum(big) / len(big)
wc = out.get("SQ_WAVE_CYCLES", 1)
print("k_ba1_fast, per launch (counters in their own units; SQ_*_CYCLES / WAIT / ACTIVE are quad-cycles summed over waves):")
for k, v in out.items(): print("%-26s %16.0f  %.4f of SQ_WAVE_CYCLES" % (k, v, v / wc))
P
find $O -name "*.db" -delete
