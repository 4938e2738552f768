#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the INDEPENDENT numpy restatement
(oracle/numpy_ref.py).  Inputs are derived from the decoded reference data files
(map_v1.npz / gt_sync.npz, made by tools/make_map_fixtures.py) and seeded generators;
outputs are what the C++ oracle (and, through it, the HIP kernels) must reproduce.

    python tools/make_golden.py          # rewrites tests/golden/golden_*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy_ref as nr  # noqa: E402
from gmmloc_amd import synth  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def cam_v1():
    f32 = lambda x: float(np.float32(x))  # camera::fx.. are float (config.h:38)
    return nr.Cam(f32(435.2046959714599), f32(435.2046959714599), f32(367.4517211914062), f32(252.2008514404297),
                  f32(47.90639384423901), 752, 480)


class CamLike:
    def __init__(self, c):
        self.__dict__.update(c.__dict__)


def main():
    m = np.load(os.path.join(G, "map_v1.npz"))
    mean, cov = m["mean"], m["cov"]
    gt = np.load(os.path.join(G, "gt_sync.npz"))
    cam, prm = cam_v1(), nr.Prm()
    comps = nr.build_components(mean, cov)

    # --- A0: component construction (subset of v1 + synthetic incl. non-degenerate) -------
    sm, sc = synth.synth_gmm(64, 7, planar_frac=0.5)
    sel = np.arange(0, mean.shape[0], 53)
    cm, cc = np.concatenate([mean[sel], sm]), np.concatenate([cov[sel], sc])
    c2 = nr.build_components(cm, cc)
    np.savez_compressed(os.path.join(G, "golden_components.npz"), mean=cm, cov=cc,
                        cov_inv=c2["cov_inv"].reshape(-1, 9), det=c2["det"], scale=c2["scale"],
                        sqrt_info=c2["sqrt_info"].reshape(-1, 9), is_deg=c2["is_deg"], is_salient=c2["is_salient"],
                        normal=c2["axis"][:, :, 0])

    # --- A1: exhaustive association on the full v1 map ------------------------------------
    pts = synth.synth_points(mean, cov, 300, 42)
    d = nr.chi2_all(mean, comps["cov_inv"], pts)
    idx = np.argmin(d, 1).astype(np.int32)
    srt = np.sort(d, 1)
    np.savez_compressed(os.path.join(G, "golden_assoc.npz"), pts=pts, idx=idx, d2=d[np.arange(300), idx],
                        margin=(srt[:, 1] - srt[:, 0]) / srt[:, 0])

    # --- A2: neighbour graph rows ---------------------------------------------------------
    rows = list(range(0, 40)) + [1000, 2000, 3298]
    nb = nr.neighbour_rows(mean, cov, comps["det"], rows)
    np.savez_compressed(os.path.join(G, "golden_nbs.npz"), rows=np.array(rows),
                        ptr=np.cumsum([0] + [len(j) for j, _ in nb]),
                        idx=np.concatenate([j for j, _ in nb]).astype(np.int32),
                        dist=np.concatenate([dd for _, dd in nb]))

    # --- A3-A5: rendered views + 2-D correspondences at ground-truth poses ----------------
    seq = gt["V1_01_easy"]
    poses = np.stack([synth.gt_row_to_Tcw(seq[i]) for i in (50, 900, 2100)])
    rng = np.random.default_rng(5)
    uv = np.stack([rng.uniform(0, 752, 250), rng.uniform(0, 480, 250)], 1)
    view_ids, view_ptr, cands, ncands = [], [0], [], []
    for p in poses:
        view = nr.render_view(mean, cov, comps, cam, p)
        view_ids += [g["id"] for g in view]
        view_ptr.append(len(view_ids))
        c, n = nr.search_correspondence(view, uv, 5)
        cands.append(c)
        ncands.append(n)
    np.savez_compressed(os.path.join(G, "golden_view.npz"), poses=poses, uv=uv, view_ids=np.array(view_ids, np.int32),
                        view_ptr=np.array(view_ptr, np.int32), cand=np.stack(cands), ncand=np.stack(ncands))

    # --- B1: optimizePoint ------------------------------------------------------------------
    deg = np.nonzero(comps["is_deg"])[0]
    N = 60
    pose = poses[0]
    T = nr.SE3.from7(pose)
    f = synth.synth_frame(mean, cov, pose, CamLike(cam), N, 77, outlier_frac=0.2, mono_frac=0.0)
    comp = f["comp"].copy()
    comp[~comps["is_deg"][comp]] = deg[0]
    pz = np.minimum(1.0, T.map(f["Xw"])[:, 2]) ** 2
    # start from a perturbed point so that Gauss-Newton has work to do
    X0 = f["Xw"] + np.random.default_rng(3).standard_normal((N, 3)) * 0.02
    res, c2p, c2s, est = [], [], [], []
    for i in range(N):
        r = nr.optimize_point(X0[i], f["obs"][i], int(f["octave"][i]), pose, comps["axis"][comp[i]][:, 0],
                              mean[comp[i]], pz[i], cam, prm)
        res.append(r[0]); c2p.append(r[1]); c2s.append(r[2]); est.append(r[3])
    np.savez_compressed(os.path.join(G, "golden_optpoint.npz"), pose=pose, pts=X0, uvr=f["obs"], octave=f["octave"],
                        comp=comp.astype(np.int32), proj_z2=pz, res=np.array(res, np.uint8), chi2_proj=np.array(c2p),
                        chi2_str=np.array(c2s), pt_est=np.stack(est))
    # --- A8: checkMapAssociation / B2: optimizeTriangulationVec (independent restatements) -------------
    rng8 = np.random.default_rng(8)
    K = mean.shape[0]
    cands = -np.ones((N, 5), np.int32)
    for i in range(N):
        n = int(rng8.integers(0, 5))
        c = [int(f["comp"][i])] + [int(x) for x in rng8.integers(0, K, 4)]
        rng8.shuffle(c)
        cands[i, :n] = c[:n]
    need = sorted(set(int(c) for c in cands.ravel() if c >= 0))
    nb_rows = dict(zip(need, [j for j, _ in nr.neighbour_rows(mean, cov, comps["det"], need)]))
    nbs = [nb_rows.get(k, np.zeros(0, int)) for k in range(K)]
    a_comp, a_pt = [], []
    for i in range(N):
        c, x = nr.check_map_association(X0[i], f["obs"][i], int(f["octave"][i]), pose, cands[i], comps, mean, nbs, cam, prm)
        a_comp.append(c); a_pt.append(x)
    np.savez_compressed(os.path.join(G, "golden_cma.npz"), pose=pose, pts=X0, uvr=f["obs"], octave=f["octave"], cand=cands,
                        ncand=(cands >= 0).sum(1).astype(np.int32), out_comp=np.array(a_comp, np.int32), out_pt=np.array(a_pt))
    pose2 = poses[1]
    T2 = nr.SE3.from7(pose2)
    pc2 = np.array([T2.map(x) for x in f["Xw"]])
    u2 = cam.fx * pc2[:, 0] / pc2[:, 2] + cam.cx + rng8.standard_normal(N) * 0.7
    v2 = cam.fy * pc2[:, 1] / pc2[:, 2] + cam.cy + rng8.standard_normal(N) * 0.7
    uvr2 = np.stack([u2, v2, np.where(rng8.uniform(size=N) < 0.5, -1.0, u2 - cam.bf / pc2[:, 2])], 1)
    uvr1 = f["obs"].copy()
    uvr1[rng8.uniform(size=N) < 0.4, 2] = -1.0
    cands2 = -np.ones((N, 5), np.int32)
    for i in range(N):
        n = int(rng8.integers(0, 4))
        cands2[i, :n] = rng8.integers(0, K, n)
    t_comp, t_pt = [], []
    for i in range(N):
        c, x = nr.optimize_triangulation(X0[i], pose, uvr1[i], int(f["octave"][i]), pose2, uvr2[i], cands[i], cands2[i],
                                         comps, mean, cam, prm)
        t_comp.append(c); t_pt.append(x)
    np.savez_compressed(os.path.join(G, "golden_tri.npz"), pose1=pose, pose2=pose2, x3d=X0, uvr1=uvr1, uvr2=uvr2,
                        oct1=f["octave"], cand1=cands, cand2=cands2, out_comp=np.array(t_comp, np.int32),
                        out_pt=np.array(t_pt))
    # --- createMapPoints per-match block: parallax test, SVD triangulation / stereo unprojection, B2, checks
    unique_normal = comps["scale"][:, 1] > 10.0 * comps["scale"][:, 0]
    NT = 150
    pA, pB = synth.gt_row_to_Tcw(seq[900]), synth.gt_row_to_Tcw(seq[915])  # neighbouring key-frames
    mt = synth.synth_tri_matches(mean, cov, pA, pB, CamLike(cam), NT, 91, allowed=unique_normal)
    c_pt, c_type, c_comp = [], [], []
    for i in range(NT):
        pt_i, ty, co = nr.create_map_point(pA, mt["uvr1"][i], mt["depth1"][i], int(mt["oct1"][i]), pB,
                                           mt["uvr2"][i], mt["depth2"][i], int(mt["oct2"][i]), mt["cand1"][i],
                                           mt["cand2"][i], comps, mean, cam, prm)
        c_pt.append(np.zeros(3) if pt_i is None else pt_i); c_type.append(ty); c_comp.append(co)
    np.savez_compressed(os.path.join(G, "golden_cmp.npz"), out_pt=np.array(c_pt), out_type=np.array(c_type, np.int32),
                        out_comp=np.array(c_comp, np.int32), **mt)

    # --- B3: optimizeCurrentPose ---------------------------------------------------------------
    out = {}
    for j, (M, seed) in enumerate([(120, 11), (60, 12), (8, 13)]):
        fr = synth.synth_frame(mean, cov, poses[j], CamLike(cam), M, seed)
        if j == 1:
            fr["octave"][::5] = -1
        p, o, n = nr.optimize_current_pose(fr["pose_init"], fr["Xw"], fr["obs"], fr["octave"], cam, prm)
        for k, v in dict(pose_init=fr["pose_init"], Xw=fr["Xw"], obs=fr["obs"], octave=fr["octave"], pose=p,
                         outlier=o, ninlier=np.array(n)).items():
            out["f%d_%s" % (j, k)] = v
    np.savez_compressed(os.path.join(G, "golden_pose.npz"), **out)

    # --- B4: jointOptimization (small problems; un-reduced dense LM vs the oracle's Schur) -----
    out = {}
    # (a) single free pose, one observation per point (the gl_track_frames shape)
    fr = synth.synth_frame(mean, cov, poses[0], CamLike(cam), 90, 21, outlier_frac=0.08)
    d = nr.chi2_all(mean, comps["cov_inv"], fr["Xw"])
    a = np.argmin(d, 1)
    assoc = np.where(d[np.arange(90), a] <= 9.0, a, -1).astype(np.int32)
    P, F, L = 1, 0, 90
    optr = np.arange(L + 1, dtype=np.int32)
    r = nr.joint_optimization(P, F, fr["pose_init"][None], [0], fr["Xw"], assoc, optr, np.zeros(L, np.int32),
                              fr["obs"], fr["octave"], comps, mean, cam, prm)
    out.update(a_P=P, a_F=F, a_poses=fr["pose_init"][None], a_prior=np.zeros(1, np.uint8), a_points=fr["Xw"],
               a_assoc=assoc, a_obs_ptr=optr, a_obs_pose=np.zeros(L, np.int32), a_obs_uvr=fr["obs"],
               a_obs_oct=fr["octave"], a_out_poses=r[0], a_out_points=r[1], a_dropped=r[2], a_erase=r[3],
               a_iters=np.array(r[4]))
    # (b) 2 free + 1 fixed poses, prior on pose 0, points seen by 2-3 poses
    rng = np.random.default_rng(9)
    base = poses[1]
    Ts = [base, synth.perturb_pose(base, rng, 0.02, 0.08), synth.perturb_pose(base, rng, 0.02, 0.08)]
    fr = synth.synth_frame(mean, cov, base, CamLike(cam), 50, 31, outlier_frac=0.0)
    L = 50
    obs_ptr, obs_pose, obs_uvr, obs_oct = [0], [], [], []
    for l in range(L):
        for pi in range(3):
            if pi == 2 and l % 3 == 0:
                continue
            Tq = nr.SE3.from7(Ts[pi])
            pc = Tq.map(fr["Xw"][l])
            octv = int(rng.integers(0, 4))
            uvr = nr.proj_stereo(pc, cam) + rng.standard_normal(3) * 1.2 ** octv * 0.7
            if (l + pi) % 4 == 0:
                uvr[2] = -1.0
            obs_pose.append(pi); obs_uvr.append(uvr); obs_oct.append(octv)
        obs_ptr.append(len(obs_pose))
    init = np.stack([synth.perturb_pose(Ts[0], rng, 0.003, 0.01), synth.perturb_pose(Ts[1], rng, 0.003, 0.01), Ts[2]])
    d = nr.chi2_all(mean, comps["cov_inv"], fr["Xw"])
    a = np.argmin(d, 1)
    assoc = np.where(d[np.arange(L), a] <= 9.0, a, -1).astype(np.int32)
    X0 = fr["Xw"] + rng.standard_normal((L, 3)) * 0.01
    r = nr.joint_optimization(2, 1, init, [1, 0], X0, assoc, np.array(obs_ptr), np.array(obs_pose),
                              np.array(obs_uvr), np.array(obs_oct), comps, mean, cam, prm)
    out.update(b_P=2, b_F=1, b_poses=init, b_prior=np.array([1, 0], np.uint8), b_points=X0, b_assoc=assoc,
               b_obs_ptr=np.array(obs_ptr, np.int32), b_obs_pose=np.array(obs_pose, np.int32),
               b_obs_uvr=np.array(obs_uvr), b_obs_oct=np.array(obs_oct, np.int32), b_out_poses=r[0],
               b_out_points=r[1], b_dropped=r[2], b_erase=r[3], b_iters=np.array(r[4]))
    np.savez_compressed(os.path.join(G, "golden_ba.npz"), **out)

    # ---- searchByProjection: three small frames through the independent (brute-force, table popcount)
    #      restatement; inputs are regenerated from the seeds by the tests, only the outputs are stored
    out = {}
    for i, (NF, NP, seed, th) in enumerate(((250, 300, 101, 3.0), (600, 500, 102, 5.0), (400, 700, 103, 1.0))):
        fr = synth.synth_match_frame(NF, NP, seed, float_uv=False)
        m, n = nr.search_by_projection(th=th, **fr)
        out["m%d_args" % i] = np.array([NF, NP, seed, th])
        out["m%d_match" % i] = m
        out["m%d_n" % i] = np.array(n)
    class CamF:  # cfg/v1.yaml intrinsics as the float config scalars
        pass
    c0 = cam_v1()
    for k in ("fx", "fy", "cx", "cy", "bf"):
        setattr(CamF, k, float(np.float32(getattr(c0, k))))
    CamF.width, CamF.height = 752, 480
    for i, (NF, NL, seed, th, motion, chk) in enumerate(((260, 220, 111, 7.0, "none", 1), (500, 420, 112, 7.0, "forward", 1),
                                                          (450, 500, 113, 14.0, "backward", 1), (300, 300, 114, 7.0, "none", 0))):
        fr = synth.synth_motion_frames(NF, NL, seed, CamF, motion, float_uv=False)
        m, n = nr.search_by_projection_frame(CamF, th=th, check_orientation=bool(chk), **fr)
        out["f%d_args" % i] = np.array([NF, NL, seed, th, {"none": 0, "forward": 1, "backward": 2}[motion], chk])
        out["f%d_match" % i] = m
        out["f%d_n" % i] = np.array(n)
    np.savez_compressed(os.path.join(G, "golden_match.npz"), **out)

    # ---- searchForTriangulation: key-frame pairs through the independent numpy restatement (inputs regenerated from the seeds)
    out = {}
    for i, (N1, N2, seed, only_stereo, chk, nodes) in enumerate(((300, 350, 201, 0, 1, 60), (700, 650, 202, 0, 1, 120), (500, 500, 203, 1, 1, 90),
                                                                 (400, 450, 204, 0, 0, 40))):
        pr = synth.synth_tri_search_pair(N1, N2, seed, CamF, n_nodes=nodes, pad=1)
        m, n = nr.search_for_triangulation(pr["kf1"], pr["kf2"], pr["fmat"], pr["epipole"], bool(only_stereo), bool(chk))
        out["t%d_args" % i] = np.array([N1, N2, seed, only_stereo, chk, nodes])
        out["t%d_match" % i] = m
        out["t%d_n" % i] = np.array(n)
    np.savez_compressed(os.path.join(G, "golden_tri_match.npz"), **out)

    # ---- searchByBoW: key-frame / frame pairs through the independent numpy restatement (inputs regenerated from the seeds)
    out = {}
    for i, (N1, N2, seed, ratio100, chk, nodes) in enumerate(((300, 350, 301, 70, 1, 60), (900, 800, 302, 70, 1, 150), (500, 520, 303, 90, 1, 25),
                                                              (400, 450, 304, 60, 0, 40))):
        kf, fr = synth.synth_bow_pair(N1, N2, seed, CamF, n_nodes=nodes)
        m, n = nr.search_by_bow(kf, fr, ratio100 / 100.0, bool(chk))
        out["b%d_args" % i] = np.array([N1, N2, seed, ratio100, chk, nodes])
        out["b%d_match" % i] = m
        out["b%d_n" % i] = np.array(n)
    np.savez_compressed(os.path.join(G, "golden_bow_match.npz"), **out)

    # ---- fuseObservations, matching half: key-frames through the independent numpy restatement (inputs regenerated from the seeds)
    out = {}
    FK = ("width", "height", "feat_uv", "feat_ur", "feat_oct", "feat_desc", "mp_uvr", "mp_level", "mp_valid", "mp_desc")
    for i, (NF, NP, seed, th10) in enumerate(((300, 260, 401, 30), (1200, 1500, 402, 30), (700, 900, 403, 50), (150, 600, 404, 25))):
        f = synth.synth_fuse_frame(NF, NP, seed)
        bi, bd, n = nr.fuse_search(*[f[k] for k in FK], th=th10 / 10.0)
        out["u%d_args" % i] = np.array([NF, NP, seed, th10])
        out["u%d_idx" % i] = bi
        out["u%d_dist" % i] = bd
        out["u%d_n" % i] = np.array(n)
    np.savez_compressed(os.path.join(G, "golden_fuse.npz"), **out)
    # ---- projection / visibility loop in front of the matchers: frames through the independent numpy restatement
    out = {}
    cam_px = cam_v1()
    for i, (NP, seed) in enumerate(((400, 501), (3000, 502), (1500, 503), (40, 504))):
        f = synth.synth_project_frame(NP, seed, cam_px)
        uvr, lvl, vc, dd, iv = nr.project_map_points(cam_px, **f)
        out["p%d_args" % i] = np.array([NP, seed])
        out["p%d_uvr" % i], out["p%d_level" % i], out["p%d_viewcos" % i], out["p%d_dist" % i], out["p%d_inview" % i] = uvr, lvl, vc, dd, iv
    np.savez_compressed(os.path.join(G, "golden_project.npz"), **out)
    print("golden vectors written to", G)


if __name__ == "__main__":
    main()
