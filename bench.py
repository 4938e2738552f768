#!/usr/bin/env python3
"""Benchmark of the gmmloc hot path on MI355X: frames/sec (associate + pose-refine),
2 000 map points x 4 096 Gaussians (BASELINE.json configs[1] shape).

A "step" = one pass of gl_track_frames over one batch of --batch synthetic frames that are
already resident in HBM: the exact fp64 Mahalanobis argmin of every frame's 2 000 map points
over the 4 096 map Gaussians (GL_ASSOC_BRUTE: cell index + same result as the N x K sweep,
gated at chi2 <= 9) + the structure-constrained refinement (single free pose,
Schur-marginalised points, 5/5/40 LM schedule).  The in-place state (poses, points) is
restored from pristine device copies inside the timed region.

    python bench.py --gpus N --steps K --warmup W [--batch B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Frames are independent units: with N > 1 every rank owns its own batch (weak scaling, GMM
replicated, no data-path collective); the timed region is bracketed by barrier +
synchronize and the MAX over ranks is reported.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from gmmloc_amd import launch  # noqa: E402

N_PTS, K_GAUSS = 2000, 4096
FLOP_PER_PAIR = 21           # centred symmetric Mahalanobis form
FLOP_PER_POINT_TRIAL = 800   # structure refine, per point per LM iteration (linearise, 3x3 inverse, Schur)
# ... and for a trial BEYOND the first of an outer iteration: g2o keeps the linearisation and only re-damps, re-solves and
# re-evaluates (3x3 inverse 40 + Schur and g 270 + transform / project / robust chi2 40, same table)
FLOP_PER_POINT_RETRIAL = 350
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X fp64 vector peak (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
PEAK_HBM_GBS = 8000.0


def make_workload(B, seed0=20200901):
    from gmmloc_amd import synth, api
    cam = api.Camera()
    mean, cov = synth.synth_gmm(K_GAUSS, 1)
    rng = np.random.default_rng(seed0)
    frames = []
    for i in range(B):
        eye = rng.uniform([-3.5, -2.5, 0.8], [2.5, 3.5, 2.2])
        tgt = eye + np.array([np.cos(rng.uniform(0, 2 * np.pi)), np.sin(rng.uniform(0, 2 * np.pi)), rng.uniform(-0.3, 0.3)]) * 3.0
        frames.append(synth.synth_frame(mean, cov, synth.look_at_pose(eye, tgt), cam, N_PTS, seed0 + i))
    return mean, cov, cam, frames


def kernel_source_fingerprint():
    """sha256 over the HIP sources the library is built from: the PMC summaries record it (tools/pmc_traffic.py), so a
    traffic figure measured on OTHER kernels than the ones being timed is flagged instead of going stale silently."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "gmmloc_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "gmmloc_amd", "csrc", "Makefile")]):
        if path.endswith((".hip", ".hpp", "Makefile")):  # (the compile flags decide spills, i.e. traffic: round 4)
            with open(path, "rb") as f:
                h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel, frames_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, collected per MI355X_MICROARCH.md in separate --pmc runs of this script;
    the traffic is per frame, so it is scaled to this run's launch size).
    None when the summary is missing: bench.py itself cannot run under two profilers.
    -> (bytes, source file, stale): stale = the summary was made from other kernel sources than the ones in the tree."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):  # the latest summary that has it
        name = os.path.basename(path)
        with open(path) as f:
            doc = json.load(f)
        t = doc.get(kernel)
        if t and "bench" in t:  # per-shape entry (the sweep: k_assoc_brute + its merge kernel)
            t = t["bench"]
        if t:
            stale = doc.get("kernel_source_sha") != kernel_source_fingerprint()
            return t["hbm_bytes_per_frame"] * frames_per_launch, "profiles/" + name, stale
    return None, None, None


def measured_counter(kernel, counter):
    """one PMC counter of `kernel` per launch, with the launch size it was measured at, from the same committed summary
    (profiles/r*_traffic.json: SQ_INSTS_VALU, SQ_WAVE_CYCLES ... per launch) -> (value, frames_per_launch) or (None, None)"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        with open(path) as f:
            doc = json.load(f)
        t = doc.get(kernel)
        if t and counter in t and t.get("frames_per_launch"):
            return float(t[counter]), int(t["frames_per_launch"])
    return None, None


_W = {}


def _cpu_one(i):
    """One frame through the oracle, SAME-MATH baseline: exhaustive all-pairs association (what the GPU path answers
    exactly), chi2 <= 9 gate, joint_optimization with one free pose.  -> seconds spent in the association."""
    orc, h, cam, frames = _W["orc"], _W["h"], _W["cam"], _W["frames"]
    f = frames[i]
    ta = time.perf_counter()
    idx, d2 = orc.associate3d(h, f["Xw"])
    ta = time.perf_counter() - ta
    assoc = np.where(d2 <= 9.0, idx, -1).astype(np.int32)
    L = f["Xw"].shape[0]
    orc.joint_optimization(h, cam, 1, 0, f["pose_init"][None].copy(), np.zeros(1, np.uint8), f["Xw"].copy(), assoc,
                           np.arange(L + 1, dtype=np.int32), np.zeros(L, np.int32), f["obs"], f["octave"])
    return ta


def _cpu_one_reference(i):
    """One frame the way the REFERENCE ALGORITHM associates (GMM::queryPoint, gaussian_mixture.cpp:545-576): exact 5-NN on
    the component means through the reference's own nanoflann kd-tree (built once per map, as the GMM constructor does)
    + chi2 of those five, best one gated at 9; then the same g2o-semantic LM (joint_optimization, one free pose)."""
    orc, h, cam, frames, tree = _W["orc"], _W["h"], _W["cam"], _W["frames"], _W["tree"]
    f = frames[i]
    ta = time.perf_counter()
    knn, _ = orc.nanoflann_tree_knn(tree, f["Xw"], 5)
    c2 = np.stack([orc.chi2(h, knn[:, j], f["Xw"]) for j in range(5)], 1)
    j = np.argmin(c2, 1)
    rows = np.arange(len(j))
    assoc = np.where(c2[rows, j] <= 9.0, knn[rows, j], -1).astype(np.int32)
    ta = time.perf_counter() - ta
    L = f["Xw"].shape[0]
    orc.joint_optimization(h, cam, 1, 0, f["pose_init"][None].copy(), np.zeros(1, np.uint8), f["Xw"].copy(), assoc,
                           np.arange(L + 1, dtype=np.int32), np.zeros(L, np.int32), f["obs"], f["octave"])
    return ta


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a box
    can expose 256 hardware threads and grant 8 of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(float(txt[0]) / float(txt[1])))))
            else:
                q = float(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = float(fh.read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_worker(seed0, budget_s):
    """Runs in a fresh process (no HIP state, so forking a pool is safe): the oracle on the first frames of the same
    workload, 1 thread - BOTH baselines of SURVEY 8d in the same run, never mixed: (i) the reference algorithm
    (nanoflann 5-NN + chi2, then LM), (ii) same-math brute force (exhaustive argmin, then LM) - then (ii) with one
    frame per worker process on all host cores.  The oracle is rebuilt on this machine with the reference's own
    flags (-O3 -march=native, gmmloc/CMakeLists.txt:7); the real reference binary cannot be built here (ROS, g2o)."""
    import multiprocessing as mp
    from tests import oracle_lib
    ncore = host_cores()
    n_all = 4 * ncore if ncore > 1 else 0
    mean, cov, cam, frames = make_workload(max(n_all, 400), seed0)
    orc = oracle_lib.load_native()
    flags = "-O3 -march=native (built on this host)"
    if orc is None:
        orc, flags = oracle_lib.load(), "-O2 -mavx2 -mfma (portable checker build: the native build failed on this host)"
    h = orc.gmm_create(mean, cov)
    _W.update(orc=orc, h=h, cam=cam, frames=frames, tree=orc.nanoflann_tree3d(mean) if (orc.nf is not None and hasattr(orc.nf, "nfref_tree3d_create")) else None)

    def timed(fn, budget):
        t0 = time.perf_counter()
        n, t_assoc = 0, 0.0
        while n < len(frames) and time.perf_counter() - t0 < budget:
            t_assoc += fn(n)
            n += 1
        return n, time.perf_counter() - t0, t_assoc

    n, dt, t_assoc = timed(_cpu_one, 0.6 * budget_s)
    ref = None
    if _W["tree"] is not None:
        nr_, dtr, tr_assoc = timed(_cpu_one_reference, 0.4 * budget_s)
        ref = {"value": nr_ / dtr, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the same workload: the reference's own nanoflann kd-tree (5-NN on the means, tree built once) "
                         "+ chi2 of the five + chi2 <= 9 gate (GMM::queryPoint), then joint_optimization (g2o-semantic LM, 1 free pose), "
                         "1 thread; association alone %.2f ms/frame" % (nr_, 1e3 * tr_assoc / nr_),
               "ms_per_frame": 1e3 * dtr / nr_, "assoc_ms_per_frame": 1e3 * tr_assoc / nr_,
               "note": "nanoflann wrapper built -O2 -mavx2 from the reference's header in the build container (the header does not "
                       "travel); it returns the best of the 5 nearest means, NOT the argmin over all Gaussians"}
    allc = None
    if n_all:
        with mp.get_context("fork").Pool(ncore) as pool:
            pool.map(_cpu_one, range(ncore))  # warm the workers
            ta = time.perf_counter()
            pool.map(_cpu_one, range(n_all), chunksize=1)
            allc = {"value": n_all / (time.perf_counter() - ta), "unit": "frames/s", "cores": ncore, "frames": n_all,
                    "how": "same-math brute force, one frame per forked worker process"}
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    brute = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
             "sample": "%d frames of the same workload (oracle = CPU port of the reference's arithmetic: all-pairs associate3d + "
                       "joint_optimization, 1 thread); association alone %.1f ms/frame" % (n, 1e3 * t_assoc / n),
             "ms_per_frame": 1e3 * dt / n, "assoc_ms_per_frame": 1e3 * t_assoc / n}
    out = dict(brute)  # the contract's cpu_baseline = the same-math baseline (what the GPU path computes, pair for pair)
    out.update({"build": flags, "cpu_model": model, "host_cores": ncore, "host_hw_threads": os.cpu_count(),
                "same_math_brute": brute, "reference_algorithm": ref, "all_cores": allc})
    print(json.dumps(out))


def cpu_baseline(seed0, budget_s=24.0):
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(seed0), str(budget_s)],
                       capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline worker failed: " + r.stderr[-2000:])
    return json.loads(r.stdout.strip().split("\n")[-1])


def stub_bench(args, ranks):
    """CPU test of the launch plumbing (tests/test_bench_contract.py): the same spawn, barrier, timing and MAX-over-ranks
    code as the real bench over gloo, with a stand-in step.  Not a measurement.  GMMLOC_STUB_STEP_MS > 0 makes the stand-in
    ASYNCHRONOUS like a kernel launch: a step only moves the time at which the stand-in device will be idle, and the rank's
    device synchronise (launch.Ranks.sync) waits for it - a clock read before that synchronise times the enqueue, not the work."""
    state = {"n": 0, "idle_at": 0.0}
    step_s = float(os.environ.get("GMMLOC_STUB_STEP_MS", "0")) / 1e3

    def step():
        state["n"] += int(np.arange(1000).sum() > 0)
        if step_s > 0:
            state["idle_at"] = max(state["idle_at"], time.perf_counter()) + step_s

    def device_sync():
        wait = state["idle_at"] - time.perf_counter()
        if wait > 0:
            time.sleep(wait)
    if step_s > 0 and ranks.backend == "gloo":
        ranks.device_sync = device_sync
    dt = launch.timed_steps(step, args.steps, args.warmup, ranks)
    total = ranks.sum(float(state["n"]))
    per_rank = ranks.gather(args.steps / max(ranks.last_own_dt, 1e-9))
    if ranks.rank == 0:
        print(json.dumps({"metric": "stub (launch plumbing test, no GPU work)", "value": total / dt, "unit": "steps/s",
                          "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                          "steps_run_all_ranks": total, "backend": ranks.backend, "ranks_in_group": ranks.group_size(),
                          "per_rank_rate": per_rank}))
    ranks.close()


def refine_flop(points, trials, outer):
    """algorithmic flop of the structure refine: an outer Levenberg iteration linearises (800 flop per point); a trial
    beyond the first of an iteration only re-damps, re-solves and re-evaluates (350)"""
    return points * (FLOP_PER_POINT_TRIAL * outer + FLOP_PER_POINT_RETRIAL * max(trials - outer, 0.0))


def refine_flop_active(edge_trials, edge_outer):
    """the same model on the edges that are really in the graph: `edge_outer` / `edge_trials` = level-0 reprojection edges summed
    over the outer iterations / over all trials (gl_ctx_set_edge_stats_buffer).  An edge the gating rounds put at level 1
    (localization_opt.cpp:799-825) is in no later linearisation and costs nothing algorithmically."""
    return FLOP_PER_POINT_TRIAL * edge_outer + FLOP_PER_POINT_RETRIAL * max(edge_trials - edge_outer, 0.0)


def extra_legs(args, ranks, ctx, gmm, cam, prm, mean, cov, pose0, Xw0, obs, octv, pose, Xw, step, start_timers, trials, outer, B, world, local, dev):
    """the legs reported beside the headline (sweep step, two streams, 1000-point class, anchored step): outside the timed region"""
    import torch
    import gmmloc_amd
    from gmmloc_amd import api
    # the same step with the association forced to the plain N x K sweep (option assoc_grid = 0): same results
    # (tests/test_gpu_track.py), the arithmetic of the same-math CPU baseline pair for pair
    sweep_steps = max(2, args.steps // 4)
    grid_before = ctx.get_option("assoc_grid")  # (may have been set through GMMLOC_ASSOC_GRID: put back what was there)
    ctx.set_option("assoc_grid", 0)
    with torch.cuda.stream(ctx.stream):
        dt_sweep = launch.timed_steps(step, sweep_steps, 1, ranks) / sweep_steps
    ctx.set_option("assoc_grid", grid_before)

    # the same step taken in turn by TWO contexts (two streams, two scratch blocks; same device, same inputs): what a host that
    # keeps two batches in flight gets - the association / set-up kernels of step k + 1 run in the tail of step k's refine,
    # where CUs are already free (a refine workgroup owns its CU's LDS and registers, so nothing else overlaps).  Reported
    # beside `value`, which stays the one-stream figure; the results of the two contexts are bit-identical.
    ctx_b = gmmloc_amd.Context(local)
    gmm_b = gmmloc_amd.GMM(ctx_b, mean, cov, prm)
    pose_b, Xw_b = pose0.clone(), Xw0.clone()
    lanes = [(ctx, gmm, pose, Xw), (ctx_b, gmm_b, pose_b, Xw_b)]
    turn = [0]

    def step2():
        c_, g_, p_, x_ = lanes[turn[0] & 1]
        turn[0] += 1
        with torch.cuda.stream(c_.stream):
            p_.copy_(pose0)
            x_.copy_(Xw0)
            gmmloc_amd.track_frames(c_, g_, cam, prm, p_, x_, obs, octv, want_d2=False)
    two_steps = max(4, 2 * (args.steps // 4))
    dt_two = launch.timed_steps(step2, two_steps, 2, ranks) / two_steps
    two_same = bool(torch.equal(pose, pose_b) and torch.equal(Xw, Xw_b))
    # the second context and its map (incl. the packed cell table) go before the legs below (nothing else holds them)
    lanes.clear()
    pose_b = Xw_b = None
    gmm_b.close()
    ctx_b.close()
    del gmm_b, ctx_b

    # outside the timed region: (a) the refine on the 1 000-point LDS class - the first 1 000 points of every frame; real
    # tracking frames have <= 1 200 features (cfg/v1.yaml:24) and two such frames share a CU, so the serial 6 x 6 solve of one
    # overlaps the passes of the other - and (b) the anchored step (gl_track_frames_anchored: prior edge on every pose)
    def refine_leg(M, prior):
        x0, ob, oc = Xw0[:, :M].contiguous(), obs[:, :M].contiguous(), octv[:, :M].contiguous()
        p, x = pose0.clone(), x0.clone()
        pr = torch.ones(B, dtype=torch.uint8, device=dev) if prior else None

        def st():
            p.copy_(pose0)
            x.copy_(x0)
            if prior:
                return gmmloc_amd.track_frames_anchored(ctx, gmm, cam, prm, p, x, ob, oc, prior=pr, want_d2=False)
            return gmmloc_amd.track_frames(ctx, gmm, cam, prm, p, x, ob, oc, want_d2=False)
        nst = max(2, args.steps // 4)
        with torch.cuda.stream(ctx.stream):
            t = launch.timed_steps(st, nst, 1, ranks, before_timed=start_timers)
        ms, nn = ctx.timing_read(api.TIMER_BA)
        ctx.timing(False)
        ntr, nou = float(trials.sum().item()), float(outer.sum().item())  # (read straight after the leg: the buffers are shared)
        ks = ms / 1e3 / max(nn, 1)
        tf = refine_flop(M, ntr, nou) / ks / 1e12 if nn else None
        return {"value": B * world * nst / t, "unit": "frames/s", "points_per_frame": M, "steps": nst, "refine_avg_launch_ms": 1e3 * ks,
                "trials_per_frame": ntr / B, "outer_iterations_per_frame": nou / B, "refine_TFLOPs": tf,
                "refine_frac_of_fp64_valu_peak": tf / PEAK_FP64_VALU_TFLOPS if tf else None}
    leg_1000 = refine_leg(1000, False)
    leg_prior = refine_leg(N_PTS, True)

    return dt_sweep, sweep_steps, dt_two, two_steps, two_same, leg_1000, leg_prior


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 without a launcher: bench.py starts them itself")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16384, help="frames per step per GPU (4 096 until round 4: the tail of a launch - CUs idle while the last frames finish - is a quarter as long at 16 384)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="the timed step only (what tools/profile_bench.sh profiles: the two-stream / sweep / class legs launch the same kernels)")
    ap.add_argument("--cpu-baseline-worker", nargs=2, metavar=("SEED", "BUDGET_S"), help=argparse.SUPPRESS)
    # tests of the launch plumbing: "gloo" = the whole path on CPU with a stand-in step; "nccl-dry" = the REAL backend selection
    # (Ranks("nccl"): set_device, init_process_group over RCCL, barrier / MAX / gather on device tensors) with the stand-in step -
    # on a GPU node a rehearsal of the N-rank job without the workload, on a CPU node it must stop at the first device call
    ap.add_argument("--stub", choices=["gloo", "nccl-dry"], default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(int(args.cpu_baseline_worker[0]), float(args.cpu_baseline_worker[1]))
        return

    # --gpus N is the number of ranks.  Under a launcher (torchrun: RANK / WORLD_SIZE set) this process is one of them;
    # started plainly with N > 1 it starts the N ranks itself (one process per GPU, RCCL) and relays rank 0's line.
    if args.gpus > 1 and not launch.is_rank():
        sys.exit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], need_gpus=args.stub != "gloo"))
    ranks = launch.Ranks("gloo" if args.stub == "gloo" else "nccl")
    if ranks.world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n"
                         % (args.gpus, ranks.world, ranks.world))
    ranks.init(always=args.stub == "nccl-dry")
    if args.stub:
        stub_bench(args, ranks)
        return

    import torch
    import gmmloc_amd
    from gmmloc_amd import api

    world, rank, local, dev = ranks.world, ranks.rank, ranks.local, ranks.device

    B = args.batch
    mean, cov, cam, frames = make_workload(B, 20200901 + 100000 * rank)
    prm = api.Params()
    ctx = gmmloc_amd.Context(local)
    gmm = gmmloc_amd.GMM(ctx, mean, cov, prm)

    def dev_t(key):
        return torch.from_numpy(np.stack([f[key] for f in frames])).to(dev)

    pose0, Xw0, obs, octv = dev_t("pose_init"), dev_t("Xw"), dev_t("obs"), dev_t("octave")
    pose, Xw = pose0.clone(), Xw0.clone()

    barrier = ranks.barrier

    trials = torch.zeros(B, dtype=torch.int32, device=dev)
    outer = torch.zeros(B, dtype=torch.int32, device=dev)
    ctx.set_stats_buffer(trials, outer)  # per-frame Levenberg trials and outer iterations of the last step (algorithmic work of k_ba1_fast)
    edges = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ctx.set_edge_stats_buffer(edges)     # ... and the level-0 reprojection edges those trials / iterations ran on

    def step():
        pose.copy_(pose0)
        Xw.copy_(Xw0)
        return gmmloc_amd.track_frames(ctx, gmm, cam, prm, pose, Xw, obs, octv, want_d2=False)

    def start_timers():
        ctx.timing(True)
        ctx.timing_read(api.TIMER_ASSOC, reset=True)
        ctx.timing_read(api.TIMER_BA, reset=True)
        ctx.timing_read(api.TIMER_BA_PREP, reset=True)

    with torch.cuda.stream(ctx.stream):
        dt = launch.timed_steps(step, args.steps, args.warmup, ranks, before_timed=start_timers)
    per_rank_rate = ranks.gather(B * args.steps / ranks.last_own_dt)  # each rank's own clock around its own steps (the headline uses the MAX)
    n_trials = float(trials.sum().item())  # (of the last timed step; the legs below reuse the buffers)
    n_outer = float(outer.sum().item())
    n_edge_trials, n_edge_outer = [float(v) for v in edges.sum(dim=0).tolist()]
    assoc_ms, assoc_n = ctx.timing_read(api.TIMER_ASSOC)
    ba_ms, ba_n = ctx.timing_read(api.TIMER_BA)
    prep_ms, prep_n = ctx.timing_read(api.TIMER_BA_PREP)
    ctx.timing(False)

    extra = not args.no_extra_legs
    dt_sweep = dt_two = None
    two_same = None
    sweep_steps = two_steps = 0
    leg_1000 = leg_prior = None
    if extra:
        dt_sweep, sweep_steps, dt_two, two_steps, two_same, leg_1000, leg_prior = extra_legs(
            args, ranks, ctx, gmm, cam, prm, mean, cov, pose0, Xw0, obs, octv, pose, Xw, step, start_timers, trials, outer, B, world, local, dev)

    # the plain N x K sweep on the same points (its roofline record), and
    # the number of chi2 evaluations the cell index needed for them
    sweep_ms = None
    idx_pairs = None
    if rank == 0:
        flat = Xw0.reshape(-1, 3)
        with torch.cuda.stream(ctx.stream):
            gmm.associate3d(flat, api.ASSOC_EXHAUSTIVE)
            ctx.timing(True)
            ctx.timing_read(api.TIMER_ASSOC, reset=True)
            for _ in range(3):
                gmm.associate3d(flat, api.ASSOC_EXHAUSTIVE)
            torch.cuda.synchronize()
            ms, nn = ctx.timing_read(api.TIMER_ASSOC, reset=True)
            ctx.timing(False)
            sweep_ms = ms / max(nn, 1)
            idx_pairs = gmm.index_work(flat)
        torch.cuda.synchronize()
        # single-frame latency (the north star also quotes a latency target): one frame per call,
        # host call -> result synchronised, on the first frame of the batch
        p1, x1, o1, c1 = pose0[:1].contiguous(), Xw0[:1].contiguous(), obs[:1].contiguous(), octv[:1].contiguous()
        pl, xl = p1.clone(), x1.clone()

        def one_frame_latency():
            lat = []
            with torch.cuda.stream(ctx.stream):
                for it in range(25):
                    pl.copy_(p1)
                    xl.copy_(x1)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    gmmloc_amd.track_frames(ctx, gmm, cam, prm, pl, xl, o1, c1, want_d2=False)
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t1)
            return 1e3 * float(np.median(lat[5:]))
        latency_ms = one_frame_latency()
        # (option ba_same_xcd = 1: the opt-in same-XCD form of the latency shape's exchange, outside the HIP memory model - gmmloc_hip.h)
        xcd_before = ctx.get_option("ba_same_xcd")
        ctx.set_option("ba_same_xcd", 1)
        latency_same_xcd_ms = one_frame_latency()
        ctx.set_option("ba_same_xcd", xcd_before)
        # host buffers in, host buffers out (what the reference host would call once per frame through the adapter):
        # page-locked staging, one enqueued copy each way, one synchronize; at the bench frame and at 700 points
        h2h = {}
        hp = api.HostFramePath(ctx, gmm, cam, prm, N_PTS)
        from gmmloc_amd import synth
        f700 = synth.synth_frame(mean, cov, synth.look_at_pose(np.array([-1.0, 0.5, 1.5]), np.array([1.5, 2.0, 1.4])), cam, 700, 20200901)
        for mpts, f0 in ((N_PTS, frames[0]), (700, f700)):
            hx, ho, hc = f0["Xw"].copy(), f0["obs"].copy(), f0["octave"].copy()
            lat = []
            for it in range(25):
                hpose, hxw = f0["pose_init"].copy(), hx.copy()
                t1 = time.perf_counter()
                hp.track_frame(hpose, hxw, ho, hc)
                lat.append(time.perf_counter() - t1)
            h2h[mpts] = 1e3 * float(np.median(lat[5:]))
        hp.close()

    if rank == 0:
        frames_total = B * args.steps * world
        pairs = float(B) * N_PTS * K_GAUSS
        assoc_s = assoc_ms / 1e3 / max(assoc_n, 1)
        sweep_s = sweep_ms / 1e3
        ach_tflops = FLOP_PER_PAIR * pairs / sweep_s / 1e12
        # algorithmic HBM bytes of one association launch: points in, records in, idx+d2 out
        alg_bytes = B * N_PTS * 24 + K_GAUSS * 96 + B * N_PTS * 12
        info = gmm.index_info()
        ba_s = ba_ms / 1e3 / max(ba_n, 1)
        ba_flop_all_points = refine_flop(N_PTS, n_trials, n_outer)    # rounds 1-4: every point priced in every trial
        ba_flop = refine_flop_active(n_edge_trials, n_edge_outer)      # round 5: the level-0 edges only
        ba_flop_every_trial = FLOP_PER_POINT_TRIAL * N_PTS * n_trials  # (what rounds 1-3 reported: every trial priced as a linearisation)
        ba_tflops = ba_flop / ba_s / 1e12 if ba_n else None
        ba_bytes = B * (N_PTS * (24 + 24 + 4 + 4 + 8 + 24 + 4) + 2 * 56)  # Xw, obs, octave, assoc, d2 in; points, final assoc out; pose in/out
        ba_traffic, ba_traffic_src, ba_traffic_stale = measured_traffic("k_ba1_fast", B)
        sw_traffic, sw_traffic_src, _ = measured_traffic("k_assoc_brute", B)
        ba_insts, ba_insts_frames = measured_counter("k_ba1_fast", "SQ_INSTS_VALU")
        insts_pt = (ba_insts * 64.0 / (ba_insts_frames * N_PTS * (n_trials / B))) if ba_insts else None
        if ba_traffic_stale:
            sys.stderr.write("bench.py: %s was measured on other kernel sources than the ones in gmmloc_amd/csrc "
                             "(roofline.traffic_stale = true): re-run tools/profile_bench.sh\n" % ba_traffic_src)
        out = {
            "metric": "frames/sec (associate+pose-refine), 2k pts x 4k GMM",
            "value": frames_total / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "ranks_in_group": ranks.group_size(),  # what the process group (RCCL) reports: == n_gpus, or the launch is not what it says
            "per_rank_frames_per_s": per_rank_rate,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if ctx.get_option("ba_step32") == 0 else "f64 (fp32-cached point step: option ba_step32)",
            "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 2000 map points x 4096 Gaussians per frame "
                                   "(95% planar map, SURVEY 8d), exact chi2 argmin over all Gaussians + structure-constrained "
                                   "refine (1 free pose, Schur, LM 5/5/40)",
                       "frames_per_step_per_gpu": B, "points_per_frame": N_PTS, "gaussians": K_GAUSS,
                       "parallelism": "frames sharded, %d rank(s)" % world},
            # dominant kernel of the step (~95 % of the time): the structure-constrained refine
            "roofline": {
                "kernel": "k_ba1_fast (single-pose LM + Schur, reprojection + point-to-plane/ellipsoid)",
                "bound": "valu_fp64",
                "achieved": ba_tflops,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (ba_tflops / PEAK_FP64_VALU_TFLOPS) if ba_tflops else None,
                "traffic": ba_traffic,
                "traffic_source": ba_traffic_src,
                "traffic_stale": ba_traffic_stale,
                "traffic_setup_kernel": measured_traffic("k_ba1_prep", B)[0],  # k_ba1_prep: gate, flags, order, normalised observations
                "avg_launch_ms": 1e3 * ba_s,
                "flop_per_launch": ba_flop,
                "units": "%d frames x (%.0f level-0 reprojection edges x outer iterations per frame x %d flop + %.0f edges x further trials "
                         "per frame x %d flop); %d points per frame, %.1f outer iterations + %.1f further trials"
                         % (B, n_edge_outer / B, FLOP_PER_POINT_TRIAL, (n_edge_trials - n_edge_outer) / B, FLOP_PER_POINT_RETRIAL,
                            N_PTS, n_outer / B, (n_trials - n_outer) / B),
                "active_edge_share": n_edge_trials / max(N_PTS * n_trials, 1.0),
                "frac_all_points_model": (ba_flop_all_points / ba_s / 1e12 / PEAK_FP64_VALU_TFLOPS) if ba_n else None,
                "frac_all_points_model_what": "rounds 1-4 priced every one of the %d points in every trial; since round 5 `frac` prices the "
                                              "level-0 reprojection edges only (edges gated out at :799-825 are in no linearisation)" % N_PTS,
                "trials_per_frame": n_trials / B,
                # VALU instructions the kernel ISSUES per point and Levenberg trial (SQ_INSTS_VALU counts wave instructions: x 64 lanes,
                # / the point-trials of the profiled launch) beside the model's FMA slots: the instruction efficiency of the trial
                "insts_per_point_trial": insts_pt,
                "insts_per_point_trial_what": "SQ_INSTS_VALU of %s x 64 / (frames x %d points x %.1f trials) of the profiled launch; the flop model "
                                              "prices a trial at %d FMA slots per point (%d for a re-trial)"
                                              % (ba_traffic_src, N_PTS, n_trials / B, FLOP_PER_POINT_TRIAL // 2, FLOP_PER_POINT_RETRIAL // 2),
                "outer_iterations_per_frame": n_outer / B,
                "frac_pricing_every_trial_as_a_linearisation": (ba_flop_every_trial / ba_s / 1e12 / PEAK_FP64_VALU_TFLOPS) if ba_n else None,
                "flop_model": "SURVEY 8d: 800 flop per point and outer iteration (linearise + 3x3 inverse + Schur + error); a Levenberg trial "
                              "beyond the first of an iteration needs no new linearisation: 350 (inverse, Schur, error).  The kernel "
                              "re-linearises in every trial (it keeps no Jacobians): that is its choice, not algorithmic work",
                "hbm": {"achieved_GBs": ba_bytes / ba_s / 1e9, "peak_GBs": PEAK_HBM_GBS,
                        "algorithmic_bytes_per_launch": ba_bytes},
            },
            # the association inside the step: exact argmin through the cell index (gather bound)
            "assoc_index": {
                "kernel": "k_assoc_cells_coop (exact cell index, wave-cooperative record gather; chi2 <= 9 resolved, the rest gated out)",
                "avg_launch_ms": 1e3 * assoc_s,
                "pairs_evaluated": idx_pairs,
                "pairs_per_point": idx_pairs / float(B * N_PTS),
                "pairs_exhaustive": pairs,
                "gather_bytes": idx_pairs * 96 + B * N_PTS * 36,
                "gather_GBs": (idx_pairs * 96 + B * N_PTS * 36) / assoc_s / 1e9 if assoc_n else None,
                "index": info,
            },
            # the plain N x K sweep (GL_ASSOC_EXHAUSTIVE) on the same points, timed outside the step
            "roofline_assoc": {
                "kernel": "k_assoc_brute (fp64 Mahalanobis argmin, all pairs)",
                "bound": "valu_fp64",
                "achieved": ach_tflops,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach_tflops / PEAK_FP64_VALU_TFLOPS,
                "traffic": sw_traffic,  # incl. the partial minima of the K-split written and merged (k_assoc_merge)
                "traffic_source": sw_traffic_src,
                "avg_launch_ms": 1e3 * sweep_s,
                "flop_per_launch": FLOP_PER_PAIR * pairs,
                "hbm": {"achieved_GBs": alg_bytes / sweep_s / 1e9, "peak_GBs": PEAK_HBM_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes},
            },
            "kernel_ms_per_step": {"associate": assoc_ms / max(args.steps, 1), "refine_setup": prep_ms / max(args.steps, 1),
                                   "refine": ba_ms / max(args.steps, 1)},
        }
        if not extra:
            print(json.dumps(out))
            ranks.close()
            return
        leg_1000["what"] = ("the same step on the first 1000 points of every frame: the 1000-point LDS class of the refine (2 frames per CU; "
                            "tracking frames have <= 1200 features, cfg/v1.yaml:24); MAX over ranks")
        out["roofline"]["class_1000"] = leg_1000
        leg_prior["what"] = ("the same step through gl_track_frames_anchored with the prior edge (EdgeSE3QuatPrior, sigma 2 deg / 1 cm) on every "
                             "frame's pose: the gauge anchor the reference's structure BA always has (localization_opt.cpp:556-581)")
        out["step_anchored_prior"] = leg_prior
        out["step_with_exhaustive_sweep"] = {
            "value": B * world / dt_sweep, "unit": "frames/s", "ms_per_step": 1e3 * dt_sweep, "steps": sweep_steps,
            "what": "the same step with GL_ASSOC_EXHAUSTIVE-style association (all 2000 x 4096 pairs per frame swept, option "
                    "assoc_grid = 0) instead of the exact cell index; MAX over ranks like the headline"}
        out["step_two_streams"] = {
            "value": B * world / dt_two, "unit": "frames/s", "ms_per_step": 1e3 * dt_two, "steps": two_steps, "bit_identical_results": two_same,
            "what": "the same step taken in turn by two contexts per GPU (two streams, two scratch blocks): the association and set-up "
                    "kernels of step k + 1 run in the tail of step k's refine; `value` above is the one-stream figure; MAX over ranks"}
        out["latency"] = {"single_frame_ms": latency_ms, "what": "gl_track_frames(B=1) call + stream sync, median of 20; default options "
                                                                  "(device-scope exchange between the frame's workgroups: the model-conforming path)",
                          "single_frame_same_xcd_ms": latency_same_xcd_ms,
                          "same_xcd_what": "the same with the opt-in option ba_same_xcd = 1 (workgroup-scope stores into the XCD's shared L2)",
                          "host_to_host_ms": h2h[N_PTS], "host_to_host_700pts_ms": h2h[700],
                          "host_to_host_what": "gl_track_frame_host: host buffers -> the context's page-locked staging -> one H2D + "
                                               "gl_track_frames + one D2H on its stream -> one synchronize -> host buffers (what the "
                                               "adapter's trackFrame calls), ctypes caller, median of 20"}
        if not args.no_cpu_baseline:
            # rank 0, after the timed region (at N > 1 the other ranks wait in the closing barrier meanwhile: the 1-thread figure
            # of the same host stays beside every line; a shorter sample there)
            cb = cpu_baseline(20200901 + 100000 * rank, 24.0 if world == 1 else 10.0)
            out["cpu_baseline"] = cb
            lat = out["latency"]
            lat["cpu_same_math_brute_ms_per_frame"] = cb["same_math_brute"]["ms_per_frame"]
            lat["speedup_vs_cpu_same_math_brute"] = cb["same_math_brute"]["ms_per_frame"] / latency_ms
            if cb.get("reference_algorithm"):
                lat["cpu_reference_algorithm_ms_per_frame"] = cb["reference_algorithm"]["ms_per_frame"]
                lat["speedup_vs_cpu_reference_algorithm"] = cb["reference_algorithm"]["ms_per_frame"] / latency_ms
            out["speedup_batched"] = {
                "vs_cpu_same_math_brute_1thread": out["step_with_exhaustive_sweep"]["value"] / cb["same_math_brute"]["value"],
                "vs_cpu_reference_algorithm_1thread": (out["value"] / cb["reference_algorithm"]["value"]) if cb.get("reference_algorithm") else None,
                "note": "like with like: the sweep step against the brute-force CPU path, the indexed step against the kd-tree CPU path"}
        print(json.dumps(out))
    ranks.close()


if __name__ == "__main__":
    main()
