"""GPU: BASELINE configs[3], the batch replay of the EuRoC sequences (gmmloc_amd/replay.py) through the HIP path at FULL
size (all 13 735 frame problems of the six sequences) on one GPU: per-frame results of the sharded, batched replay are
bit-identical to one unsharded call (both refine kernels add in one canonical order whatever the batch); an oracle leg
holds evenly spaced frames of every sequence to the CPU oracle (optimizeCurrentPose, then the anchored structure refine =
joint_optimization(P = 1, prior)) at the north-star tolerance; the TUM trajectories of the library's writer parse back and
score against gt_sync.  The RCCL collectives of the N > 1 path are exercised with one rank (all_gather / all_reduce on the
GPU); N > 1 itself needs the driver's 8-GPU node (tools/replay_euroc.py --gpus N, gmmloc_amd/launch.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import api, replay, traj
from tests.conftest import GOLDEN, ROOT
from tests.test_gpu_pose import pose_err
from tests.test_gpu_anchor import oracle_anchored

pytestmark = pytest.mark.gpu


def test_replay_euroc_full_size_matches_unsharded_oracle_and_scores(gpu, oracle, tmp_path):
    torch, ctx = gpu
    cam, prm = api.Camera(), api.Params()
    maps, frames = replay.materialise_euroc(GOLDEN, cam, M=300)
    assert len(frames) == 13735 and {f["map"] for f in frames} == {"map_v1", "map_v2"}
    gmms = {name: gmmloc_amd.GMM(ctx, mean, cov, prm) for name, (mean, cov) in maps.items()}
    compute = replay.TrackCompute(ctx, gmms, cam, prm, anchor="prior")  # the prior edge on the tracker's pose
    whole = compute(frames)  # every frame of a map in ONE call
    # the replay proper: batches of 1024, and the same with the frames dealt to 3 "ranks" (run one after the other)
    res1, _ = replay.replay(frames, compute, 0, 1, None, "cpu", batch=1024)
    res3 = np.zeros_like(res1)
    for r in range(3):
        idx = replay.shard_indices(len(frames), r, 3)
        res3[idx] = np.concatenate([compute([frames[i] for i in idx[s:s + 700]]) for s in range(0, len(idx), 700)])
    for res in (res1, res3):
        assert np.array_equal(res[:, :17], whole[:, :17])  # poses and counters: the same bits whatever the sharding / batching

    # ---- oracle leg: 8 evenly spaced frames of every sequence --------------------------------------------------------
    hs = {name: oracle.gmm_create(mean, cov) for name, (mean, cov) in maps.items()}
    checked = 0
    for seq, mapname in replay.EUROC_SEQUENCES:
        sel = [i for i, f in enumerate(frames) if f["seq"] == seq]
        for i in sel[::max(1, len(sel) // 8)][:8]:
            f = frames[i]
            p_ref, o_ref, n_ref = oracle.optimize_current_pose(cam, f["pose_init"], f["Xw"], f["obs"], f["octave"])
            dt, dr = pose_err(res1[i, :7], p_ref)
            assert dt < 1e-6 and dr < 1e-6 and int(res1[i, 14]) == n_ref, (seq, i, dt, dr)
            # the outlier mask of that frame: the HIP call on the frame alone (same bits as inside the replay)
            T = lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).cuda()
            pose = T(f["pose_init"])
            outl, _ = api.optimize_current_pose(ctx, cam, prm, pose, T(f["Xw"]), T(f["obs"]), T(f["octave"]))
            assert np.array_equal(outl.cpu().numpy()[0], o_ref) and np.array_equal(pose.cpu().numpy()[0], res1[i, :7])
            # "Discard outliers" (tracking.cpp:313-324), then the anchored structure refine from the tracker's pose
            g = dict(f, pose_init=p_ref, octave=np.where(o_ref != 0, -1, f["octave"]).astype(np.int32))
            keep, ps_ref, _, a_ref, _ = oracle_anchored(oracle, hs[mapname], cam, g, True, 0)
            dt, dr = pose_err(res1[i, 7:14], ps_ref)
            assert dt < 1e-6 and dr < 1e-6, (seq, i, dt, dr)
            assert int(res1[i, 15]) == int((a_ref >= 0).sum()) and int(res1[i, 16]) == len(keep)
            checked += 1
    assert checked == 48
    for h in hs.values():
        oracle.gmm_destroy(h)

    rep = replay.score_euroc(frames, res1, str(tmp_path))
    assert set(rep) == {s for s, _ in replay.EUROC_SEQUENCES}
    for seq, r in rep.items():
        st, xyz, quat = traj.read_tum(r["tum"])
        assert len(st) >= 2 and np.all(np.diff(st) > 0) and np.allclose(np.linalg.norm(quat, axis=1), 1.0, atol=1e-8)
        assert r["associated_per_frame"] > 100 and r["inliers_per_frame"] > 200
        # optimizeCurrentPose pulls the 2 cm / 0.6 deg perturbed poses to the generating trajectory (4 - 5 mm)
        assert r["ape_rmse_m"] < 0.01 and r["ape_rmse_m"] < 0.5 * r["ape_rmse_initial_m"], (seq, r)
        # the structure refine frees the points: with its own stereo observation alone a point carries no information about
        # the pose beyond its Gaussian edge, so on EXACT map points the pose cannot beat the tracker's; the prior edge
        # (sigma 1 cm / 2 deg) holds it at 16 - 19 mm (4 - 6 cm without an anchor, the round before)
        assert r["ape_rmse_structure_m"] < 0.03, (seq, r)


@pytest.mark.parametrize("map_sigma", [0.0, 0.02])
def test_replay_with_fixed_observers(gpu, map_sigma):
    """The reference's regime: its structure BA always has fixed observer key-frames (localization_opt.cpp:491-516).
    With two fixed observers per frame (the ground-truth poses 10 and 20 rows earlier, own noisy observations) and the
    prior edge the structure-refined pose scores < 1 cm on exact map points, and on a realistic local map (2 cm of noise
    on the points the tracker is given) it is BETTER than the tracker's pose on all six sequences."""
    torch, ctx = gpu
    cam, prm = api.Camera(), api.Params()
    maps, frames = replay.materialise_euroc(GOLDEN, cam, M=300, limit=300, map_sigma=map_sigma, fixed=2)
    gmms = {name: gmmloc_amd.GMM(ctx, mean, cov, prm) for name, (mean, cov) in maps.items()}
    compute = replay.TrackCompute(ctx, gmms, cam, prm, anchor="fixed")
    res, _ = replay.replay(frames, compute, 0, 1, None, "cpu", batch=512)
    rep = replay.score_euroc(frames, res)
    for seq, r in rep.items():
        if map_sigma == 0.0:
            assert r["ape_rmse_m"] < 0.01 and r["ape_rmse_structure_m"] < 0.01, (seq, r)
        else:
            assert r["ape_rmse_structure_m"] < r["ape_rmse_m"] < 0.03, (seq, r)


def test_replay_cli_with_rccl_collectives_at_world_1():
    """tools/replay_euroc.py end to end in its own process with the nccl (= RCCL) process group of ONE rank: the
    all_gather of the result rows and the all_reduce of the timing run on the GPU, as they do for N > 1."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_euroc.py"), "--gpus", "1", "--limit", "40", "--batch", "64",
                          "--collective-at-world-1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.split("\n") if l.startswith("{")][-1])  # RCCL prints its banner on stdout too
    assert d["frames"] == 240 and d["backend"] == "nccl" and d["frames_per_s"] > 0 and d["anchor"] == "prior"
    assert all(v["ape_rmse_m"] < 0.01 and v["ape_rmse_structure_m"] < 0.03 for v in d["sequences"].values())
