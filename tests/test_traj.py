"""CPU: TUM trajectory writer (Map::summarize, map.cpp:162-188) and APE evaluation (evo_euroc.py:28-57)."""
import numpy as np

from gmmloc_amd import traj, synth


def test_tum_writer_format_and_roundtrip(tmp_path, gt_sync):
    gt = gt_sync["V1_01_easy"][:50]
    # T_wc from the ground-truth rows (T_cw inverted): q -> conj(q), t -> -R^T t
    pose_wc = []
    for r in gt:
        T = synth.gt_row_to_Tcw(r)
        R = synth.quat_to_R(T[:4])
        pose_wc.append(np.concatenate([[-T[0], -T[1], -T[2], T[3]], -R.T @ T[4:]]))
    pose_wc = np.stack(pose_wc)
    stamps = 1403715273.262142976 + 0.05 * np.arange(50)
    p = tmp_path / "traj_est.txt"
    traj.write_tum(p, stamps, pose_wc)
    lines = p.read_text().split("\n")
    assert lines[-1] == "" and len(lines) == 51
    # std::fixed, setprecision(6) stamp, setprecision(9) pose, single spaces
    want = "%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f" % (stamps[0], *pose_wc[0][4:], *pose_wc[0][:4])
    assert lines[0] == want
    t, xyz, quat = traj.read_tum(p)
    np.testing.assert_allclose(t, stamps, atol=1e-6)
    np.testing.assert_allclose(xyz, pose_wc[:, 4:], atol=1e-9)
    np.testing.assert_allclose(quat, pose_wc[:, :4], atol=1e-9)


def test_umeyama_recovers_similarity_and_ape():
    rng = np.random.default_rng(1)
    N = 400
    gt = np.cumsum(rng.standard_normal((N, 3)) * 0.05, 0)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    s, t = 1.3, np.array([2.0, -1.0, 0.5])
    est = ((gt - t) @ R) / s  # gt = s R est + t
    s2, R2, t2 = traj.umeyama(est, gt)
    assert abs(s2 - s) < 1e-12 and np.abs(R2 - R).max() < 1e-12 and np.abs(t2 - t).max() < 1e-12
    stamps = np.arange(N) * 0.05
    # estimate sampled at every second stamp, shifted by 2 ms, with 1 cm noise
    noise = rng.standard_normal((N // 2, 3)) * 0.01
    r = traj.ape_translation(stamps, gt, stamps[::2] + 0.002, est[::2] + noise / s)
    assert r["n"] == N // 2 and abs(r["scale"] - s) < 0.02
    assert 0.010 < r["rmse"] < 0.025 and r["mean"] <= r["rmse"] <= r["max"]
    # stamps further apart than max_diff are not associated
    r2 = traj.ape_translation(stamps, gt, stamps[::2] + 0.02, est[::2])
    assert r2["n"] == 0 and np.isnan(r2["rmse"])
