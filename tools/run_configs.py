#!/usr/bin/env python3
"""Measure the BASELINE.json configs that are not the bench.py headline.
Prints one JSON line per config; run on the GPU box:

    python tools/run_configs.py [--configs 2,3,4,5] [--frames-cap N]

 config 2  synthetic 2 000 points x 4 096 Gaussians, association only: batched pairs/s,
           single-frame latency, CPU same-math brute force and CPU reference algorithm
           (the reference's own nanoflann 5-NN + Mahalanobis, i.e. GMM::queryPoint's search)
 config 3  V1_03-shaped replay (2 149 frames synthesised from the real v1 map + GT poses,
           M ~ U{150..1200}): search2d + optimizeCurrentPose + associate/structure-BA
 config 4  all six sequences (13 735 frames) through gl_track_frames on this rank's shard
 config 5  stress: 50 000 points x 65 536 Gaussians association, fp64 VALU roofline
No EuRoC imagery exists here: frame problems are synthetic-from-real-map and labelled so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 78.6


def ev_time(torch, fn, reps, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def config2(torch, ctx, out):
    import gmmloc_amd
    from gmmloc_amd import synth
    from tests import oracle_lib
    orc = oracle_lib.load()
    res = {"config": "2: synthetic 2000 pts x 4096 Gaussians, association only"}
    for seed in (1, 2, 3):
        mean, cov = synth.synth_gmm(4096, seed)
        g = gmmloc_amd.GMM(ctx, mean, cov)
        pts1 = torch.from_numpy(synth.synth_points(mean, cov, 2000, seed)).cuda()
        ptsB = torch.from_numpy(synth.synth_points(mean, cov, 2000 * 512, seed + 50)).cuda()
        from gmmloc_amd import api
        EX = api.ASSOC_EXHAUSTIVE
        t1 = ev_time(torch, lambda: g.associate3d(pts1, EX), 200, ctx.stream)
        tB = ev_time(torch, lambda: g.associate3d(ptsB, EX), 10, ctx.stream)
        t1i = ev_time(torch, lambda: g.associate3d(pts1), 200, ctx.stream)
        tBi = ev_time(torch, lambda: g.associate3d(ptsB), 10, ctx.stream)
        res["seed%d" % seed] = {"sweep_single_frame_latency_us": 1e6 * t1, "sweep_batched_pairs_per_s": 2000 * 512 * 4096 / tB,
                                "sweep_batched_tflops_algorithmic": 21 * 2000 * 512 * 4096 / tB / 1e12,
                                "sweep_single_frame_tflops_algorithmic": 21 * 2000 * 4096 / t1 / 1e12,
                                "index_single_frame_latency_us": 1e6 * t1i, "index_batched_points_per_s": 2000 * 512 / tBi,
                                "index_pairs_per_point": g.index_work(ptsB) / (2000.0 * 512),
                                "index_unresolved_frac": float((g.associate3d(ptsB)[1] > 9.000009).float().mean().item())}
        if seed == 1:
            h = orc.gmm_create(mean, cov)
            p = pts1.cpu().numpy()
            t0 = time.perf_counter()
            for _ in range(20):
                orc.associate3d(h, p)
            tb = (time.perf_counter() - t0) / 20
            res["cpu_same_math_brute_1thread"] = {"ms_per_frame": 1e3 * tb, "pairs_per_s": 2000 * 4096 / tb}
            if orc.nf is not None:
                t0 = time.perf_counter()
                for _ in range(20):
                    ki, kd, kc = orc.nanoflann_knn(mean, p, 5)
                    orc.chi2(h, ki[:, 0].copy(), p)
                tk = (time.perf_counter() - t0) / 20
                res["cpu_reference_algorithm_nanoflann_1thread"] = {"ms_per_frame": 1e3 * tk,
                                                                    "note": "kd-tree build + 5-NN + chi2 (queryPoint)"}
            orc.gmm_destroy(h)
    out(res)


def v1_frames(seq, n_cap, Mlo, Mhi, seed0, sig_scale=1.0):
    from gmmloc_amd import synth, api
    cam = api.Camera()
    d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz"))
    mean, cov = d["mean"], d["cov"]
    gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))[seq]
    rng = np.random.default_rng(seed0)
    n = min(n_cap, gt.shape[0])
    frames = []
    for i in range(n):
        M = int(rng.integers(Mlo, Mhi + 1))
        f = synth.synth_frame(mean, cov, synth.gt_row_to_Tcw(gt[i]), cam, M, 20200901 + i)
        frames.append(f)
    return mean, cov, cam, frames


def pad_batch(torch, frames, Mmax):
    B = len(frames)
    pose = np.stack([f["pose_init"] for f in frames])
    Xw = np.zeros((B, Mmax, 3))
    obs = np.zeros((B, Mmax, 3))
    octv = -np.ones((B, Mmax), np.int32)
    for b, f in enumerate(frames):
        m = f["Xw"].shape[0]
        Xw[b, :m], obs[b, :m], octv[b, :m] = f["Xw"], f["obs"], f["octave"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return T(pose), T(Xw), T(obs), T(octv)


def config3(torch, ctx, out, cap):
    import gmmloc_amd
    from gmmloc_amd import api
    mean, cov, cam, frames = v1_frames("V1_03_difficult", cap, 150, 1200, 3)
    prm = api.Params()
    g = gmmloc_amd.GMM(ctx, mean, cov, prm)
    pose, Xw, obs, octv = pad_batch(torch, frames, 1200)
    uv = obs[:, :, :2].contiguous()
    nfeat = (octv >= 0).sum(1).to(torch.int32)
    B = len(frames)
    t_s2d = ev_time(torch, lambda: g.search2d(cam, pose, uv, nfeat, 5), 2, ctx.stream)
    t_pose = ev_time(torch, lambda: gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose.clone(), Xw, obs, octv), 3, ctx.stream)
    t_trk = ev_time(torch, lambda: gmmloc_amd.track_frames(ctx, g, cam, prm, pose.clone(), Xw.clone(), obs, octv, False), 3, ctx.stream)
    # the same frames grouped by size on the host (<= 500 / <= 1000 / rest), each group padded to its
    # own maximum: the refine then runs 4 / 2 / 1 frames per CU
    groups = [[f for f in frames if lo < f["Xw"].shape[0] <= hi] for lo, hi in ((0, 500), (500, 1000), (1000, 1 << 30))]
    packed = [pad_batch(torch, gfr, max(f["Xw"].shape[0] for f in gfr)) for gfr in groups if gfr]

    def run_groups():
        for gp, gx, go, gc in packed:
            gmmloc_amd.track_frames(ctx, g, cam, prm, gp.clone(), gx.clone(), go, gc, False)
    t_trk_g = ev_time(torch, run_groups, 3, ctx.stream)
    out({"config": "3: V1_03-shaped replay, synthetic-from-real-map (v1.gmm K=3299 + gt_sync poses), M~U{150..1200}",
         "associate3d+structureBA_grouped_by_size_frames_per_s": B / t_trk_g,
         "frames": B,
         "search2d_renderView+searchCorrespondence_frames_per_s": B / t_s2d,
         "optimizeCurrentPose_4x10LM_frames_per_s": B / t_pose,
         "associate3d+structureBA_frames_per_s": B / t_trk,
         "ms_per_frame": {"search2d": 1e3 * t_s2d / B, "pose": 1e3 * t_pose / B, "track": 1e3 * t_trk / B}})


def config4(torch, ctx, out, cap):
    import gmmloc_amd
    from gmmloc_amd import api, replay
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    tot, tsum = 0, 0.0
    per = {}
    for seq, mp in (("V1_01_easy", "v1"), ("V1_02_medium", "v1"), ("V1_03_difficult", "v1")):
        mean, cov, cam, frames = v1_frames(seq, cap, 300, 300, 4)
        frames = [frames[i] for i in replay.shard_indices(len(frames), rank, world)]
        prm = api.Params()
        g = gmmloc_amd.GMM(ctx, mean, cov, prm)
        pose, Xw, obs, octv = pad_batch(torch, frames, 300)
        t = ev_time(torch, lambda: gmmloc_amd.track_frames(ctx, g, cam, prm, pose.clone(), Xw.clone(), obs, octv, False), 3, ctx.stream)
        per[seq] = len(frames) / t
        tot += len(frames)
        tsum += t
    out({"config": "4: batch replay of the V1 sequences (config-1 frame shape: M=300 stereo observations, synthetic-from-real-map), "
                   "this rank's round-robin shard", "rank": rank, "world": world, "frames": tot,
         "frames_per_s": tot / tsum, "per_sequence_frames_per_s": per,
         "note": "V2 sequences use v2.gmm identically; frames of different sequences are independent units"})


def config5(torch, ctx, out):
    import gmmloc_amd
    from gmmloc_amd import synth
    mean, cov = synth.synth_gmm(65536, 5)
    t0 = time.perf_counter()
    g = gmmloc_amd.GMM(ctx, mean, cov)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    pts = torch.from_numpy(synth.synth_points(mean, cov, 50000, 5)).cuda()
    from gmmloc_amd import api
    t = ev_time(torch, lambda: g.associate3d(pts, api.ASSOC_EXHAUSTIVE), 5, ctx.stream)
    ti = ev_time(torch, lambda: g.associate3d(pts), 5, ctx.stream)
    pairs = 50000.0 * 65536
    out({"config": "5: stress 50 000 pts x 65 536 Gaussians, association (fp64, exact)", "ms": 1e3 * t,
         "index_ms": 1e3 * ti, "index": g.index_info(), "index_pairs_per_point": g.index_work(pts) / 50000.0,
         "index_unresolved_frac": float((g.associate3d(pts)[1] > 9.000009).float().mean().item()),
         "pairs_per_s": pairs / t, "tflops_algorithmic_21_per_pair": 21 * pairs / t / 1e12,
         "frac_of_fp64_valu_peak": 21 * pairs / t / 1e12 / PEAK,
         "algorithmic_hbm_bytes": 50000 * 36 + 65536 * 96, "hbm_GBs": (50000 * 36 + 65536 * 96) / t / 1e9,
         "map_build_s_incl_65536^2_neighbour_graph": t_build, "neighbour_edges": int(g.lib.gl_gmm_nbs_count(g.h))})


def config_ba(torch, ctx, out):
    """Local BA (gl_joint_optimization) timing on synthetic multi-view problems over the real v1 map."""
    import gmmloc_amd
    from gmmloc_amd import api
    from tests.test_gpu_ba import make_ba_problem
    d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz"))
    mean, cov = d["mean"], d["cov"]
    gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_01_easy"]
    cam, prm = api.Camera(), api.Params()
    g = gmmloc_amd.GMM(ctx, mean, cov, prm)
    from tests import oracle_lib
    orc = oracle_lib.load()
    hor = orc.gmm_create(mean, cov)
    cpu_ms = {}
    res = {"config": "local BA (jointOptimization), synthetic multi-view problems on v1.gmm",
           "flop_model": "per Levenberg trial: observations x 337 (P1 linearise 225: transform, residual, Huber, w J^T J, R^T A R; P3 112: "
                         "back-substitution term + error at the trial state) + points x 150 (3x3 inverse, GMM edge, step) + (pose pairs per point, "
                         "diagonal included) x 324 (G1^T [A1 R1 D^-1 R2^T A2] G2 = 162 FMA) + diagonal pairs x 150 + 2 n^3 / 3 (LDL^T, n = 6 P); "
                         "level-0 counts (an upper bound after the gating); trials counted by the kernel (gl_ctx_set_stats_buffer)",
           "peak_TFLOPs": 78.6}
    for (P, F, L, B) in ((8, 4, 1500, 1), (8, 4, 1500, 64), (8, 4, 1500, 256), (20, 8, 3000, 1)):
        probs = [make_ba_problem(mean, cov, gt, cam, P, F, L, 100 + b) for b in range(min(B, 4))]
        probs = [probs[b % len(probs)] for b in range(B)]
        NOBS = max(len(p["obs_pose"]) for p in probs)
        pad = lambda a, n: np.concatenate([a, np.zeros((n - len(a),) + a.shape[1:], a.dtype)])
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        idx, d2 = g.associate3d(T(np.concatenate([p["points"] for p in probs])))
        assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(B, L).contiguous()
        poses0, pts0 = T(np.stack([p["poses"] for p in probs])), T(np.stack([p["points"] for p in probs]))
        prior, optr = T(np.stack([p["prior"] for p in probs])), T(np.stack([p["obs_ptr"] for p in probs]))
        opose = T(np.stack([pad(p["obs_pose"], NOBS) for p in probs]))
        ouvr = T(np.stack([pad(p["obs_uvr"], NOBS) for p in probs]))
        ooct = T(np.stack([pad(p["obs_oct"], NOBS) for p in probs]))
        trials = torch.zeros(B, dtype=torch.int32, device="cuda")
        ctx.set_stats_buffer(trials)
        t = ev_time(torch, lambda: api.joint_optimization(ctx, g, cam, prm, P, F, poses0.clone(), prior, pts0.clone(), assoc, optr, opose, ouvr, ooct), 3, ctx.stream)
        torch.cuda.synchronize()
        ctx.set_stats_buffer(None)
        if (P, F, L) not in cpu_ms:  # the oracle (CPU port of the reference's arithmetic, 1 thread) on the window's first problem
            a0 = assoc[0].cpu().numpy()
            t0 = time.perf_counter()
            orc.joint_optimization(hor, cam, P, F, probs[0]["poses"], probs[0]["prior"], probs[0]["points"], a0, probs[0]["obs_ptr"], probs[0]["obs_pose"],
                                   probs[0]["obs_uvr"], probs[0]["obs_oct"])
            cpu_ms[(P, F, L)] = 1e3 * (time.perf_counter() - t0)
        flop = 0.0
        tr = trials.cpu().numpy()
        for b in range(B):
            p = probs[b]
            nfree = np.array([(p["obs_pose"][p["obs_ptr"][l]:p["obs_ptr"][l + 1]] < P).sum() for l in range(L)])
            pairs, diag = float((nfree * (nfree + 1) // 2).sum()), float(nfree.sum())
            per_trial = len(p["obs_pose"]) * 337.0 + L * 150.0 + pairs * 324.0 + diag * 150.0 + 2.0 * (6 * P) ** 3 / 3.0
            flop += per_trial * float(tr[b])
        res["P%d_F%d_L%d_B%d" % (P, F, L, B)] = {"ms_per_call": 1e3 * t, "ms_per_problem": 1e3 * t / B, "observations": int(NOBS),
                                                 "trials_per_problem": float(tr.mean()),
                                                 "cpu_oracle_1thread_ms_per_problem": cpu_ms[(P, F, L)],
                                                 "speedup_vs_cpu_oracle_1thread": cpu_ms[(P, F, L)] / (1e3 * t / B),
                                                 "us_per_trial": 1e6 * t / max(float(tr.mean()), 1.0),
                                                 "roofline_ba": {"kernel": "kp_lin + kp_schur + kp_assemble + kp_solve + kp_trial (pipelined shape)" if (B <= 8 and NOBS >= 5000) else "k_ba_gen",
                                                                 "bound": "latency (valu_fp64 peak for reference)", "flop_per_launch": flop,
                                                                 "achieved": flop / t / 1e12, "peak": 78.6, "unit": "TFLOP/s", "frac": flop / t / 1e12 / 78.6}}
    orc.gmm_destroy(hor)
    out(res)


def config_kf(torch, ctx, out):
    """Key-frame association chain (A7 + A8: search2d -> checkMapAssociation) and the batched point
    refinements (B1 optimizePoint, B2 optimizeTriangulationVec) on the real v1 map."""
    import gmmloc_amd
    from gmmloc_amd import api, synth
    d = np.load(os.path.join(ROOT, "tests", "golden", "map_v1.npz"))
    mean, cov = d["mean"], d["cov"]
    gt = np.load(os.path.join(ROOT, "tests", "golden", "gt_sync.npz"))["V1_02_medium"]
    cam, prm = api.Camera(), api.Params()
    g = gmmloc_amd.GMM(ctx, mean, cov, prm)
    B, N = 128, 1000
    fr = [synth.synth_frame(mean, cov, synth.gt_row_to_Tcw(gt[(11 + i * 13) % gt.shape[0]]), cam, N, 50 + i,
                            mono_frac=0.0, outlier_frac=0.1) for i in range(B)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    poses = T(np.stack([f["pose_gt"] for f in fr]))
    pts0 = T(np.stack([f["Xw"] for f in fr]) + np.random.default_rng(2).standard_normal((B, N, 3)) * 0.02)
    uvr = T(np.stack([f["obs"] for f in fr]))
    octv = T(np.stack([f["octave"] for f in fr]))
    uv = uvr[:, :, :2].contiguous()
    cand, ncand, _, _ = g.search2d(cam, poses, uv, None, k=5)
    t_s2d = ev_time(torch, lambda: g.search2d(cam, poses, uv, None, k=5), 3, ctx.stream)
    t_cma = ev_time(torch, lambda: api.check_map_association(ctx, g, cam, prm, poses, pts0.clone(), uvr, octv, cand, ncand), 3, ctx.stream)
    # B1 on B*N independent problems (component = first candidate, else a degenerate one)
    flags = g.get(api.F_FLAGS)
    deg0 = int(np.nonzero(flags & 1)[0][3])
    comp = cand[:, :, 0].reshape(-1).clone()
    comp[comp < 0] = deg0
    comp = torch.where(torch.from_numpy((flags & 1).astype(bool)).cuda()[comp.long()], comp, torch.full_like(comp, deg0))
    poseN = poses[:, None, :].expand(B, N, 7).reshape(-1, 7).contiguous()
    pz2 = torch.ones(B * N, dtype=torch.float64, device="cuda")
    t_b1 = ev_time(torch, lambda: api.optimize_point(ctx, g, cam, prm, pts0.reshape(-1, 3), uvr.reshape(-1, 3), octv.reshape(-1),
                                                     poseN, comp.to(torch.int32), pz2), 3, ctx.stream)
    # B2: pair frame b with frame b+1 (same points seen from two poses is not needed for timing: the
    # kernel cost is the candidates x 20 GN iterations)
    M2 = (B // 2) * N
    p1 = poses[0::2][:, None, :].expand(B // 2, N, 7).reshape(-1, 7).contiguous()
    p2 = poses[1::2][:, None, :].expand(B // 2, N, 7).reshape(-1, 7).contiguous()
    x3 = pts0[0::2].reshape(-1, 3).clone()
    t_b2 = ev_time(torch, lambda: api.optimize_triangulation(ctx, g, cam, prm, x3.clone(), p1, uvr[0::2].reshape(-1, 3).contiguous(),
                                                             octv[0::2].reshape(-1).contiguous(), p2, uvr[1::2].reshape(-1, 3).contiguous(),
                                                             octv[1::2].reshape(-1).contiguous(), cand[0::2].reshape(M2, 5).contiguous(),
                                                             ncand[0::2].reshape(-1).contiguous(), cand[1::2].reshape(M2, 5).contiguous(),
                                                             ncand[1::2].reshape(-1).contiguous()), 3, ctx.stream)
    # one key-frame at a time (the reference's caller)
    p1k, x1k, u1k, o1k = poses[:1].contiguous(), pts0[:1].contiguous(), uvr[:1].contiguous(), octv[:1].contiguous()
    c1k, n1k = g.search2d(cam, p1k, uv[:1].contiguous(), None, k=5)[:2]
    t_cma1 = ev_time(torch, lambda: api.check_map_association(ctx, g, cam, prm, p1k, x1k.clone(), u1k, o1k, c1k, n1k), 5, ctx.stream)
    t_b21 = ev_time(torch, lambda: api.optimize_triangulation(ctx, g, cam, prm, x3[:N].clone(), p1[:N].contiguous(), uvr[0].contiguous(),
                                                              octv[0].contiguous(), p2[:N].contiguous(), uvr[1].contiguous(), octv[1].contiguous(),
                                                              cand[0].contiguous(), ncand[0].contiguous(), cand[1].contiguous(),
                                                              ncand[1].contiguous()), 5, ctx.stream)
    out({"config": "key-frame chain on v1.gmm: %d key-frames x %d features" % (B, N),
         "single_keyframe_checkMapAssociation_us": 1e6 * t_cma1, "single_keyframe_optimizeTriangulation_1000_matches_us": 1e6 * t_b21,
         "search2d_keyframes_per_s": B / t_s2d, "checkMapAssociation_keyframes_per_s": B / t_cma,
         "checkMapAssociation_features_per_s": B * N / t_cma, "mean_candidates_per_feature": float(ncand.float().mean().item()),
         "optimizePoint_problems_per_s": B * N / t_b1, "optimizeTriangulation_problems_per_s": M2 / t_b2})


def config_match(torch, ctx, out):
    """searchByProjection on synthetic ORB-like frames: 1 200 features x 1 500 projected
    map points (a local map), th = 3."""
    import gmmloc_amd
    from gmmloc_amd import api, synth
    from tests import oracle_lib
    from tests.test_gpu_match import KEYS
    orc = oracle_lib.load()
    NF, NP, B = 1200, 1500, 2048
    uniq = [synth.synth_match_frame(NF, NP, 500 + b) for b in range(64)]
    frames = [uniq[b % 64] for b in range(B)]
    cam = api.Camera()
    cam.width, cam.height = 752, 480
    T = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in frames]))).cuda()
    a = {k: T(k) for k in KEYS}
    a["mp_level"] = a["mp_level"].to(torch.int32)
    args = [a[k] for k in KEYS]
    t = ev_time(torch, lambda: api.search_by_projection(ctx, cam, *args, th=3.0), 5, ctx.stream)
    t1 = ev_time(torch, lambda: api.search_by_projection(ctx, cam, *[x[:1] for x in args], th=3.0), 50, ctx.stream)
    t0 = time.perf_counter()
    for f in uniq:
        orc.search_by_projection(th=3.0, **f)
    tc = (time.perf_counter() - t0) / len(uniq)
    bytes_frame = NF * (16 + 4 + 4 + 32 + 1 + 4) + NP * (24 + 4 + 8 + 1 + 32)
    # frame-to-frame overload (trackWithMotionModel): 1 200 features vs the 1 000 map points of the last frame
    from tests.test_gpu_match import FKEYS, CamF
    NL = 1000
    uq = [synth.synth_motion_frames(NF, NL, 900 + b, CamF, "none") for b in range(64)]
    fr2 = [uq[b % 64] for b in range(B)]
    a2 = [torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in fr2]))).cuda() for k in FKEYS]
    t2 = ev_time(torch, lambda: api.search_by_projection_frame(ctx, api.Camera(), *a2, th=7.0), 5, ctx.stream)
    t0 = time.perf_counter()
    for f in uq:
        orc.search_by_projection_frame(CamF, th=7.0, **f)
    tc2 = (time.perf_counter() - t0) / len(uq)
    # searchForTriangulation: key-frame pairs of 1 200 features, DBoW2 feature vectors of ~240 nodes (level 4 of a 10^6-word tree
    # gives a few hundred nodes per key-frame)
    from tests.test_gpu_match import _pack_pairs
    up = [synth.synth_tri_search_pair(1200, 1200, 1300 + b, api.Camera(), n_nodes=240) for b in range(64)]
    k1, k2, fm, ep = _pack_pairs(torch, [up[b % 64] for b in range(B)])
    t3 = ev_time(torch, lambda: api.search_for_triangulation(ctx, k1, k2, fm, ep, False, True), 5, ctx.stream)
    one = _pack_pairs(torch, up[:1])
    t3l = ev_time(torch, lambda: api.search_for_triangulation(ctx, *one, False, True), 50, ctx.stream)
    t0 = time.perf_counter()
    for p in up:
        orc.search_for_triangulation(p["kf1"], p["kf2"], p["fmat"], p["epipole"], False, True)
    tc3 = (time.perf_counter() - t0) / len(up)
    bytes_pair = 2 * 1200 * (16 + 4 + 4 + 4 + 32 + 1 + 4) + 2 * 240 * 8 + 1200 * 4
    out({"config": "searchForTriangulation: %d key-frame pairs x 1200 + 1200 features, ~240 vocabulary nodes each" % B,
         "pairs_per_s": B / t3, "single_pair_latency_us": 1e6 * t3l, "algorithmic_bytes_per_pair": bytes_pair,
         "algorithmic_GBs": B * bytes_pair / t3 / 1e9, "cpu_oracle_1thread_pairs_per_s": 1.0 / tc3})
    # searchByBoW (trackReferenceKeyFrame): reference key-frame of 1 200 features, 70 % of them with a map point, against a frame of 1 200
    from tests.test_gpu_match import _pack_bow
    ub = [synth.synth_bow_pair(1200, 1200, 1500 + b, api.Camera(), n_nodes=240) for b in range(64)]
    bk, bf = _pack_bow(torch, [ub[b % 64] for b in range(B)])
    t4 = ev_time(torch, lambda: api.search_by_bow(ctx, bk, bf, 0.7, True), 5, ctx.stream)
    one_b = _pack_bow(torch, ub[:1])
    t4l = ev_time(torch, lambda: api.search_by_bow(ctx, *one_b, 0.7, True), 50, ctx.stream)
    t0 = time.perf_counter()
    for p in ub:
        orc.search_by_bow(p[0], p[1], 0.7, True)
    tc4 = (time.perf_counter() - t0) / len(ub)
    bytes_bow = 2 * 1200 * (4 + 32 + 4) + 1200 + 2 * 240 * 8 + 1200 * 4
    out({"config": "searchByBoW: %d key-frame / frame pairs x 1200 + 1200 features, ~240 vocabulary nodes each" % B,
         "pairs_per_s": B / t4, "single_pair_latency_us": 1e6 * t4l, "algorithmic_bytes_per_pair": bytes_bow,
         "algorithmic_GBs": B * bytes_bow / t4 / 1e9, "cpu_oracle_1thread_pairs_per_s": 1.0 / tc4})
    # the projection / visibility loop in front of the matcher, and the two as Tracking::searchLocalPoints in one call: 1 200 features,
    # a local map of 3 000 points of which a fifth to a quarter is in view
    from tests.test_gpu_match import _pack_project
    PK = ("pose_cw", "t_wc", "pos", "normal", "max_dist", "min_dist", "cand")
    NPL = 3000
    ul = [synth.synth_local_points_frame(NF, NPL, 1700 + b, api.Camera()) for b in range(64)]
    fl = [ul[b % 64] for b in range(B)]
    proj = _pack_project(torch, fl)
    TL = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in fl]))).cuda()
    feat = [TL(k) for k in ("feat_uv", "feat_ur", "feat_oct", "feat_desc", "feat_taken")]
    mpd = TL("mp_desc")
    t5 = ev_time(torch, lambda: api.project_map_points(ctx, api.Camera(), *proj), 5, ctx.stream)
    t6 = ev_time(torch, lambda: api.search_local_points(ctx, api.Camera(), *feat, *proj, mpd, th=3.0), 5, ctx.stream)
    t6l = ev_time(torch, lambda: api.search_local_points(ctx, api.Camera(), *[x[:1] for x in feat], *[x[:1] for x in proj], mpd[:1], th=3.0), 50, ctx.stream)
    t0 = time.perf_counter()
    for f in ul:
        orc.project_map_points(api.Camera(), **{k: f[k] for k in PK})
    tc5 = (time.perf_counter() - t0) / len(ul)
    t0 = time.perf_counter()
    for f in ul:
        uvr, lvl, vc, dd, iv, n = orc.project_map_points(api.Camera(), **{k: f[k] for k in PK})
        orc.search_by_projection(752, 480, f["feat_uv"], f["feat_ur"], f["feat_oct"], f["feat_desc"], f["feat_taken"], uvr, lvl, vc, iv, f["mp_desc"], th=3.0)
    tc6 = (time.perf_counter() - t0) / len(ul)
    bytes_proj = NPL * (57 + 45) + 80
    out({"config": "projection / visibility loop (Frame::project3 + checkScaleAndVisible): %d frames x %d map points" % (B, NPL),
         "frames_per_s": B / t5, "points_per_s": B * NPL / t5, "algorithmic_bytes_per_frame": bytes_proj, "algorithmic_GBs": B * bytes_proj / t5 / 1e9,
         "cpu_oracle_1thread_frames_per_s": 1.0 / tc5})
    out({"config": "searchLocalPoints (projection loop + searchByProjection in one call): %d frames x %d features x %d map points, th=3" % (B, NF, NPL),
         "frames_per_s": B / t6, "single_frame_latency_us": 1e6 * t6l, "cpu_oracle_1thread_frames_per_s": 1.0 / tc6})
    # fuseObservations, the matching half: 1 200 features of a neighbour key-frame x 1 500 projected map points
    from tests.test_gpu_match import _pack_fuse, FUSE_KEYS
    uf = [synth.synth_fuse_frame(NF, NP, 1900 + b, float_coords=True) for b in range(64)]
    fa = _pack_fuse(torch, [uf[b % 64] for b in range(B)])
    t7 = ev_time(torch, lambda: api.fuse_search(ctx, cam, *fa, th=3.0), 5, ctx.stream)
    t0 = time.perf_counter()
    for f in uf:
        orc.fuse_search(f["width"], f["height"], *[f[k] for k in FUSE_KEYS], th=3.0)
    tc7 = (time.perf_counter() - t0) / len(uf)
    out({"config": "fuseObservations, matching half: %d key-frames x %d features x %d map points, th=3" % (B, NF, NP),
         "frames_per_s": B / t7, "cpu_oracle_1thread_frames_per_s": 1.0 / tc7})
    out({"config": "searchByProjection(CurrentFrame, LastFrame): %d frame pairs x %d features x %d last-frame map points, th=7" % (B, NF, NL),
         "frames_per_s": B / t2, "cpu_oracle_1thread_frames_per_s": 1.0 / tc2})
    out({"config": "searchByProjection: %d frames x %d features x %d map points, th=3" % (B, NF, NP),
         "frames_per_s": B / t, "single_frame_latency_us": 1e6 * t1, "algorithmic_bytes_per_frame": bytes_frame,
         "algorithmic_GBs": B * bytes_frame / t / 1e9, "cpu_oracle_1thread_frames_per_s": 1.0 / tc})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--frames-cap", type=int, default=100000)
    a = ap.parse_args()
    import torch
    import gmmloc_amd
    ctx = gmmloc_amd.Context(int(os.environ.get("LOCAL_RANK", "0")))
    out = lambda d: print(json.dumps(d), flush=True)
    for c in a.configs.split(","):
        {"2": lambda: config2(torch, ctx, out), "3": lambda: config3(torch, ctx, out, a.frames_cap),
         "4": lambda: config4(torch, ctx, out, a.frames_cap), "5": lambda: config5(torch, ctx, out),
         "ba": lambda: config_ba(torch, ctx, out), "kf": lambda: config_kf(torch, ctx, out),
         "match": lambda: config_match(torch, ctx, out)}[c]()


if __name__ == "__main__":
    main()
