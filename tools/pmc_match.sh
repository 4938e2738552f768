#!/bin/bash
# On the GPU box: counters under the matcher-side kernels (tools/match_legs.py), one rocprofv3 --pmc pass per counter group
# (--kernel-trace only beside --pmc: gpurun refuses the other trace domains with counters):
#     bash tools/pmc_match.sh <tag> [legs]      -> gpurun_out/<tag>_match_traffic.json, gpurun_out/<tag>_match_legs.jsonl
set -u
TAG=${1:-r5}
LEGS=${2:-proj,frame,tri,bow,fuse}
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/pmc_match_$TAG; rm -rf $O; mkdir -p $O
python tools/match_legs.py --legs $LEGS --reps 20 > gpurun_out/${TAG}_match_legs.jsonl 2> $O/legs.err
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CNT -d $O/pmc$i -o pmc -- python tools/match_legs.py --legs $LEGS --reps 2 > $O/pmc$i.out 2> $O/pmc$i.err
done
python - $O gpurun_out/${TAG}_match_legs.jsonl <<'P' > gpurun_out/${TAG}_match_traffic.json
import glob, json, os, sqlite3, sys, collections
root, legs_path = sys.argv[1], sys.argv[2]
legs = [json.loads(l) for l in open(legs_path) if l.startswith("{")]
out = {}
for db in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*_results.db"), recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"))
    except sqlite3.Error as e:
        print("cannot read", db, e, file=sys.stderr)
        continue
    per = collections.defaultdict(float)
    for name, cn, v, disp in rows:
        full = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = full.split("(")[0]
        if not any(k in short for k in ("k_search", "k_fuse", "k_project")):
            continue
        per[(short, cn, disp)] += v
    by = collections.defaultdict(list)
    for (k, cn, disp), v in per.items():
        by[(k, cn)].append(v)
    for (k, cn), vals in by.items():
        vals.sort(reverse=True)
        big = [v for v in vals if v >= 0.5 * vals[0]] if vals[0] > 0 else vals   # the batch launches (the warm-up launch included)
        out.setdefault(k, {})[cn] = sum(big) / len(big)
        out[k][cn + "_launches"] = len(big)
for k, d in out.items():
    leg = next((l for l in legs if l["kernel"] in k), None)
    if leg:
        d["leg"] = leg
        alg = leg["algorithmic_bytes_per_unit"] * leg["B"]
        d["algorithmic_bytes_per_launch"] = alg
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # FETCH_SIZE counts 64-byte requests in KB; on gfx950 a request is 128 B (MI355X_MICROARCH.md): x 2
        d["hbm_bytes_per_launch_corrected"] = d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024
        if leg:
            d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch_corrected"] / alg
            d["hbm_GBs_at_measured_time"] = d["hbm_bytes_per_launch_corrected"] / (leg["ms_per_launch"] * 1e-3) / 1e9
    if d.get("SQ_WAVE_CYCLES"):
        d["valu_active_over_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
        d["wait_any_over_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_over_lds_active"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
out["how"] = ("rocprofv3 --kernel-trace --pmc <group> -- python tools/match_legs.py --reps 2, one run per counter group (tools/pmc_match.sh); "
              "means over the batch launches of each kernel; times and rounds per unit from the un-profiled run of the same script")
print(json.dumps(out, indent=1))
P
find $O -name "*.db" -delete
cat gpurun_out/${TAG}_match_traffic.json | head -120
