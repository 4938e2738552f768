"""Per-frame half of the randomised parity soak (tools/soak.py), with the CPU side - generation of the round's inputs and the
oracle's single-pose structure refine - spread over a process pool, so that the 86 000 rounds of round 3 (21 500 frames per
map, each refined plain and with the prior edge) re-run in minutes instead of an hour.  Same rounds (= seeds), same checks, same
strict tolerances as tools/soak.py: gl_track_frames and gl_track_frames_anchored against the oracle (pose 1e-6 m / 1e-6 rad,
associations and chi2 equal), batch shape against latency shape (equal bits).  A deviation is classified on the spot by the
oracle's own sensitivity (tools/soak_classify.py: 12 re-orderings + 36 one-ulp perturbations of the observations): a frame on
which the oracle itself moves by more than 1e-6 is ill-conditioned; one on which it does not is a REAL deviation.
    python tools/soak_track.py [rounds] [--start R] [--procs N] [--maps map_v1,map_v2]"""
import argparse
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

_W = {}


def _init():
    from gmmloc_amd import api
    from tests import oracle_lib
    from tools import soak_cases as sc
    _W.update(sc=sc, orc=oracle_lib.load(), cam=api.Camera(), gts=sc.load_gt(), maps={})


def _map(mapname):
    if mapname not in _W["maps"]:
        mean, cov = _W["sc"].load_map(mapname)
        _W["maps"][mapname] = (mean, cov, _W["orc"].gmm_create(mean, cov))
    return _W["maps"][mapname]


def work(job):
    mapname, r = job
    sc, orc, cam = _W["sc"], _W["orc"], _W["cam"]
    mean, cov, h = _map(mapname)
    f = sc.gen(mapname, r, mean, cov, _W["gts"], cam)["track"]
    keep, p_ref, _, a_ref, idx0, d20 = sc.track_oracle(orc, h, cam, f)
    _, pp_ref, _, ap_ref, _, _ = sc.track_oracle(orc, h, cam, f, prior=True)
    return dict(mapname=mapname, r=r, f={k: f[k] for k in ("pose_init", "Xw", "obs", "octave")}, keep=keep, p_ref=p_ref, a_ref=a_ref,
                idx0=idx0, d20=d20, pp_ref=pp_ref, ap_ref=ap_ref)


def probe(job):
    """the oracle's own sensitivity on one frame (soak_classify.classify_track)"""
    from tests.test_gpu_pose import pose_err
    mapname, r, prior = job
    sc, orc, cam = _W["sc"], _W["orc"], _W["cam"]
    mean, cov, h = _map(mapname)
    f = sc.gen(mapname, r, mean, cov, _W["gts"], cam)["track"]
    keep, p_ref, _, _, _, _ = sc.track_oracle(orc, h, cam, f, prior=prior)
    rng = np.random.default_rng(0)
    d = []
    for _ in range(12):
        _, p1, _, _, _, _ = sc.track_oracle(orc, h, cam, f, rng.permutation(len(keep)), prior=prior)
        d.append(max(pose_err(p1, p_ref)))
    for _ in range(36):
        g = dict(f)
        g["obs"] = f["obs"] * (1 + 3e-16 * rng.standard_normal(f["obs"].shape))
        g["obs"][f["obs"] < 0] = f["obs"][f["obs"] < 0]
        _, p1, _, _, _, _ = sc.track_oracle(orc, h, cam, g, prior=prior)
        d.append(max(pose_err(p1, p_ref)))
    d = np.array(d)
    return dict(mapname=mapname, r=r, prior=prior, median=float(np.median(d)), max=float(d.max()), above=int((d > 1e-6).sum()), n=len(d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rounds", nargs="?", type=int, default=2000)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--maps", default="map_v1,map_v2")
    args = ap.parse_args()
    ctxm = mp.get_context("spawn")  # (the parent holds a HIP context: no fork)
    pool = ctxm.Pool(args.procs, initializer=_init)
    import torch
    import gmmloc_amd
    from gmmloc_amd import api
    from tests.test_gpu_pose import pose_err
    from tools import soak_cases as sc
    ctx = gmmloc_amd.Context(0)
    cam, prm = api.Camera(), api.Params()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    one = torch.ones(1, dtype=torch.uint8).cuda()
    t0 = time.time()
    checked = dict(track=0, track_prior=0, shape=0)
    devs = []
    for mapname in args.maps.split(","):
        mean, cov = sc.load_map(mapname)
        g = gmmloc_amd.GMM(ctx, mean, cov, prm)
        first = args.start + (-args.start) % 4
        jobs = [(mapname, r) for r in range(first, args.start + args.rounds, 4)]  # (soak.py refines every 4th round's frame)
        for w in pool.imap(work, jobs, chunksize=8):
            r, f = w["r"], w["f"]
            res = {}
            for kind in ("track", "track_prior"):
                out = []
                for shape in (-1, 0):
                    ctx.set_option("ba_shape", shape)
                    pose, Xw = T(f["pose_init"][None]), T(f["Xw"][None])
                    if kind == "track":
                        assoc, d2 = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, T(f["obs"][None]), T(f["octave"][None]))
                    else:
                        assoc, d2 = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose, Xw, T(f["obs"][None]), T(f["octave"][None]), prior=one)[:2]
                    torch.cuda.synchronize()
                    out.append((pose, Xw, assoc))
                ctx.set_option("ba_shape", -1)
                checked["shape"] += 1
                if not all(torch.equal(x, y) for x, y in zip(out[0], out[1])):
                    devs.append(dict(kind="shape", mapname=mapname, r=r, what=kind))
                    print("DEVIATION shape %s round %d (%s): batch and latency shape differ in bits" % (mapname, r, kind), flush=True)
                res[kind] = (out[0][0].cpu().numpy()[0], out[0][2].cpu().numpy()[0], d2.cpu().numpy()[0])
            keep = w["keep"]
            for kind, p_ref, a_ref in (("track", w["p_ref"], w["a_ref"]), ("track_prior", w["pp_ref"], w["ap_ref"])):
                pose, assoc, d2 = res[kind]
                dt, dr = pose_err(pose, p_ref)
                a_ok = bool(np.array_equal(assoc[keep], a_ref))
                d_ok = bool(np.array_equal(d2[keep], w["d20"])) if kind == "track" else True
                checked[kind] += 1
                if not (dt < 1e-6 and dr < 1e-6 and a_ok and d_ok):
                    devs.append(dict(kind=kind, mapname=mapname, r=r, M=len(f["octave"]), dt=dt, dr=dr, a_ok=a_ok, d_ok=d_ok))
                    print("DEVIATION %-11s %s round %d  M %d pose |dt| %.3g m |dr| %.3g rad, associations equal %s, chi2 equal %s"
                          % (kind, mapname, r, len(f["octave"]), dt, dr, a_ok, d_ok), flush=True)
        del g
    t1 = time.time()
    # classification: the oracle's own sensitivity on every deviating frame
    todo = [(d["mapname"], d["r"], d["kind"] == "track_prior") for d in devs if d["kind"] != "shape"]
    real = 0
    for d, p in zip([d for d in devs if d["kind"] != "shape"], pool.map(probe, todo, chunksize=1)):
        stable = p["above"] == 0
        real += int(stable or not (d["a_ok"] and d["d_ok"]))
        print("CLASSIFIED %-11s %s r%d M=%d hip_vs_oracle=(%.3g, %.3g) decisions_equal=%s oracle_probes=%d oracle_probe_median=%.3g oracle_probe_max=%.3g "
              "oracle_probes_above_1e-6=%d -> %s" % (d["kind"], d["mapname"], d["r"], d["M"], d["dt"], d["dr"], d["a_ok"] and d["d_ok"], p["n"], p["median"], p["max"],
                                                    p["above"], "REAL (the oracle is stable here)" if stable else "ill-conditioned (the oracle itself moves)"), flush=True)
    pool.close()
    nshape = sum(d["kind"] == "shape" for d in devs)
    print("soak_track: rounds %d .. %d per map; checked %s; deviations %d (+ %d shape), of which on a stable oracle or with different decisions: %d; "
          "%.0f s GPU + oracle, %.0f s classification, %d processes"
          % (args.start, args.start + args.rounds, checked, len(devs) - nshape, nshape, real, t1 - t0, time.time() - t1, args.procs))
    sys.exit(1 if real or nshape else 0)


if __name__ == "__main__":
    main()
