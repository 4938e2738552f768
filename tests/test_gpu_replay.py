"""GPU: BASELINE configs[3], the batch replay of the EuRoC sequences (gmmloc_amd/replay.py) run through the HIP path
at world = 1 on a subset: per-frame results of the sharded, batched replay are bit-identical to one unsharded call
(the refine adds in one canonical order whatever the batch), the TUM trajectories of the library's writer parse back,
and the refined poses score an APE against gt_sync far below the perturbed initial poses'.  The RCCL collectives of
the N > 1 path are exercised with one rank (all_gather / all_reduce on the GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import api, replay, traj
from tests.conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_replay_euroc_subset_matches_unsharded_and_scores(gpu, tmp_path):
    torch, ctx = gpu
    cam, prm = api.Camera(), api.Params()
    maps, frames = replay.materialise_euroc(GOLDEN, cam, M=300, limit=48)
    assert len(frames) == 6 * 48 and {f["map"] for f in frames} == {"map_v1", "map_v2"}
    gmms = {name: gmmloc_amd.GMM(ctx, mean, cov, prm) for name, (mean, cov) in maps.items()}
    compute = replay.TrackCompute(ctx, gmms, cam, prm)
    whole = compute(frames)  # every frame of a map in ONE call
    # the replay proper: batches of 32, and the same with the frames dealt to 3 "ranks" (run one after the other)
    res1, _ = replay.replay(frames, compute, 0, 1, None, "cpu", batch=32)
    res3 = np.zeros_like(res1)
    for r in range(3):
        idx = replay.shard_indices(len(frames), r, 3)
        res3[idx] = np.concatenate([compute([frames[i] for i in idx[s:s + 20]]) for s in range(0, len(idx), 20)])
    for res in (res1, res3):
        assert np.array_equal(res[:, :17], whole[:, :17])  # poses and counters: the same bits whatever the sharding / batching
    rep = replay.score_euroc(frames, res1, str(tmp_path))
    assert set(rep) == {s for s, _ in replay.EUROC_SEQUENCES}
    for seq, r in rep.items():
        st, xyz, quat = traj.read_tum(r["tum"])
        assert len(st) >= 2 and np.all(np.diff(st) > 0) and np.allclose(np.linalg.norm(quat, axis=1), 1.0, atol=1e-8)
        assert r["associated_per_frame"] > 100 and r["inliers_per_frame"] > 200
        # optimizeCurrentPose pulls the 2 cm / 0.6 deg perturbed poses to the generating trajectory (4 mm with the oracle);
        # the structure refine frees the points as well and has no gauge anchor: reported, loosely bounded
        assert r["ape_rmse_m"] < 0.01 and r["ape_rmse_m"] < 0.5 * r["ape_rmse_initial_m"], (seq, r)
        assert r["ape_rmse_structure_m"] < 0.2, (seq, r)


def test_replay_cli_with_rccl_collectives_at_world_1():
    """tools/replay_euroc.py end to end in its own process with the nccl (= RCCL) process group of ONE rank: the
    all_gather of the result rows and the all_reduce of the timing run on the GPU, as they do for N > 1."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_euroc.py"), "--limit", "40", "--batch", "64",
                          "--collective-at-world-1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.split("\n") if l.startswith("{")][-1])  # RCCL prints its banner on stdout too
    assert d["frames"] == 240 and d["backend"] == "nccl" and d["frames_per_s"] > 0
    assert all(v["ape_rmse_m"] < 0.01 for v in d["sequences"].values())
