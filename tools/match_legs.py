#!/usr/bin/env python3
"""The matcher-side kernels alone, on the shapes of tools/run_configs.py's `match` legs - what tools/pmc_match.sh puts under
rocprofv3 (no oracle, no CPU timing: a few launches per kernel):
    python tools/match_legs.py [--legs proj,frame,tri,bow,fuse] [--reps 3] [--B 2048] [--prof]
Prints one JSON line per leg: HIP-event time per launch, units/s, the algorithmic bytes per unit, and the mean number of rounds
of the owner fixed point per unit (GL_COUNTER_MATCH_ROUNDS / _UNITS).  --prof: the library was built with -DGL_MATCH_PROF
(tools/build_variant.sh matchprof "-DGL_MATCH_PROF" gl_match.hip; GMMLOC_HIP_LIB=...): k_search_by_projection then leaves
{grid build, rounds, output} clocks (s_memtime / 16) and the round count in feat_match[0..3] of every frame."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# algorithmic bytes per unit (= what tools/run_configs.py prices): every input array once, the outputs once
NF, NP, NL = int(os.environ.get("MATCH_LEGS_NF", "1200")), int(os.environ.get("MATCH_LEGS_NP", "1500")), 1000
BYTES = {
    "proj": NF * (16 + 4 + 4 + 32 + 1 + 4) + NP * (24 + 4 + 8 + 1 + 32),
    "frame": NF * (16 + 4 + 4 + 4 + 32 + 1 + 4) + NL * (24 + 1 + 4 + 4 + 32) + 112,
    "tri": 2 * 1200 * (16 + 4 + 4 + 4 + 32 + 1 + 4) + 2 * 240 * 8 + 1200 * 4,
    "bow": 2 * 1200 * (4 + 32 + 4) + 1200 + 2 * 240 * 8 + 1200 * 4,
    "fuse": NF * (16 + 4 + 4 + 32) + NP * (24 + 4 + 1 + 32 + 8) + NP * 8,
}
KERNEL = {"proj": "k_search_by_projection<0", "frame": "k_search_by_projection<1", "tri": "k_search_for_triangulation",
          "bow": "k_search_by_bow", "fuse": "k_fuse_search"}


def ev_time(torch, fn, reps, stream):
    for _ in range(40):  # (a fresh process starts at idle clocks: these launches are too short to ramp them inside the timed ones)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / reps


def build_legs(torch, ctx, B, legs):
    from gmmloc_amd import api, synth
    from tests.test_gpu_match import KEYS, FKEYS, CamF, _pack_pairs, _pack_bow, _pack_fuse
    out = {}
    cam = api.Camera()
    cam.width, cam.height = 752, 480
    U = min(64, B)
    if "proj" in legs:
        uniq = [synth.synth_match_frame(NF, NP, 500 + b) for b in range(U)]
        a = {k: torch.from_numpy(np.ascontiguousarray(np.stack([uniq[b % U][k] for b in range(B)]))).cuda() for k in KEYS}
        a["mp_level"] = a["mp_level"].to(torch.int32)
        args = [a[k] for k in KEYS]
        out["proj"] = (lambda: api.search_by_projection(ctx, cam, *args, th=3.0), args)
    if "frame" in legs:
        uq = [synth.synth_motion_frames(NF, NL, 900 + b, CamF, "none") for b in range(U)]
        a2 = [torch.from_numpy(np.ascontiguousarray(np.stack([uq[b % U][k] for b in range(B)]))).cuda() for k in FKEYS]
        out["frame"] = (lambda: api.search_by_projection_frame(ctx, api.Camera(), *a2, th=7.0), a2)
    if "tri" in legs:
        up = [synth.synth_tri_search_pair(1200, 1200, 1300 + b, api.Camera(), n_nodes=240) for b in range(U)]
        k1, k2, fm, ep = _pack_pairs(torch, [up[b % U] for b in range(B)])
        out["tri"] = (lambda: api.search_for_triangulation(ctx, k1, k2, fm, ep, False, True), None)
    if "bow" in legs:
        ub = [synth.synth_bow_pair(1200, 1200, 1500 + b, api.Camera(), n_nodes=240) for b in range(U)]
        bk, bf = _pack_bow(torch, [ub[b % U] for b in range(B)])
        out["bow"] = (lambda: api.search_by_bow(ctx, bk, bf, 0.7, True), None)
    if "fuse" in legs:
        uf = [synth.synth_fuse_frame(NF, NP, 1900 + b, float_coords=True) for b in range(U)]
        fa = _pack_fuse(torch, [uf[b % U] for b in range(B)])
        out["fuse"] = (lambda: api.fuse_search(ctx, cam, *fa, th=3.0), None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="proj,frame,tri,bow,fuse")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--B", type=int, default=2048)
    ap.add_argument("--prof", action="store_true")
    a = ap.parse_args()
    import torch
    import gmmloc_amd
    from gmmloc_amd import api
    ctx = gmmloc_amd.Context(0)
    legs = build_legs(torch, ctx, a.B, a.legs.split(","))
    for name, (fn, args) in legs.items():
        ctx.counter_read(api.COUNTER_MATCH_ROUNDS)
        ctx.counter_read(api.COUNTER_MATCH_UNITS)
        t = ev_time(torch, fn, a.reps, ctx.stream)
        rounds, units = ctx.counter_read(api.COUNTER_MATCH_ROUNDS), ctx.counter_read(api.COUNTER_MATCH_UNITS)
        rec = {"leg": name, "kernel": KERNEL[name], "B": a.B, "ms_per_launch": 1e3 * t, "units_per_s": a.B / t,
               "algorithmic_bytes_per_unit": BYTES[name], "algorithmic_GBs": a.B * BYTES[name] / t / 1e9,
               "fixed_point_rounds_per_unit": (rounds / units) if units else None}
        if a.prof and name in ("proj", "frame"):
            with torch.cuda.stream(ctx.stream):
                res = fn()
            torch.cuda.synchronize()
            fm = res[0] if isinstance(res, (tuple, list)) else res
            w = fm[:, :17].cpu().numpy().astype(np.float64)
            rec["prof_walks_of_thread_0"] = {"setup_x16": float(w[:, 7].mean()), "loop_x16": float(w[:, 8].mean()), "of_loop_flush_x16": float(w[:, 9].mean()),
                                            "iterations": float(w[:, 10].mean()), "flushes": float(w[:, 11].mean()), "walks": float(w[:, 12].mean()), "listed_in_last_round": float(w[:, 13].mean()), "later_rounds_records_x16": float(w[:, 14].mean()), "later_rounds_listed_walks_x16": float(w[:, 15].mean()), "later_rounds_compare_x16": float(w[:, 16].mean())}
            rec["prof_mean_clocks_x16"] = {"grid_build": float(w[:, 0].mean()), "rounds": float(w[:, 1].mean()),
                                          "output": float(w[:, 2].mean()), "n_rounds": float(w[:, 3].mean()),
                                          "n_rounds_max": float(w[:, 3].max()), "of_rounds_class_sort": float(w[:, 4].mean()),
                                          "of_rounds_round_1": float(w[:, 5].mean())}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
