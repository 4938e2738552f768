#!/usr/bin/env python3
"""Per-kernel PMC sums of the rocprofv3 --pmc passes tools/profile_bench.sh made (rocpd SQLite, one db per pass) ->
JSON: counters averaged per launch of the bench launches (16 384 frames since round 5), and the HBM-side traffic
(FETCH_SIZE x 2 on gfx950, see /opt/skills/guides/MI355X_MICROARCH.md: the counter is in KB of 64-B requests that are
128 B on this chip) + WRITE_SIZE, per launch and per frame.
    python tools/pmc_traffic.py gpurun_out/prof_r2 > profiles/history/r2_traffic.json"""
import glob
import json
import os
import sqlite3
import sys

FRAMES = 16384  # frames per launch of the default bench step (4 096 until round 4)


def main(root):
    out = {}
    for db in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*_results.db"), recursive=True)):
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, counter_name, value, dispatch_id, grid_size from counters_collection"))
        except sqlite3.Error:
            try:
                rows = [(a, b, v, d, 0) for a, b, v, d in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection")]
            except sqlite3.Error as e:
                print("cannot read", db, e, file=sys.stderr)
                continue
        per = {}
        for name, cn, v, disp, grid in rows:
            full = name.replace("(anonymous namespace)::", "").split("(")[0]
            short = full.split("::")[-1]
            if short == "k_ba1_fast":  # one entry per instance (LDS class / latency shape / anchored): bafd2000::k_ba1_fast ...
                short = "::".join(full.split("::")[-2:])
            per.setdefault((short, cn, disp), 0.0)
            per[(short, cn, disp)] += v
        # keep the big launches of each kernel (the B = 1 latency launches of bench.py are far smaller): the top third by value
        bykc = {}
        for (k, cn, disp), v in per.items():
            bykc.setdefault((k, cn), []).append(v)
        for (k, cn), vals in bykc.items():
            vals.sort(reverse=True)
            big = [v for v in vals if v >= 0.5 * vals[0]] if vals[0] > 0 else vals
            out.setdefault(k, {})[cn] = sum(big) / len(big)
            out[k][cn + "_launches"] = len(big)
    if "bafd2000::k_ba1_fast" in out:
        out["k_ba1_fast"] = dict(out["bafd2000::k_ba1_fast"], instance="bafd2000::k_ba1_fast (the bench's 2 000-point frames, plain refine)")
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch_corrected"] = d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024
            d["hbm_bytes_per_frame"] = d["hbm_bytes_per_launch_corrected"] / FRAMES
            d["frames_per_launch"] = FRAMES
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
            d["valu_active_over_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
            d["wait_any_over_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["kernel_source_sha"] = bench.kernel_source_fingerprint()  # bench.py flags a summary made from other sources as stale
    out["how"] = ("rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline, one run per "
                  "counter group (tools/profile_bench.sh); averages over the bench launches")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
