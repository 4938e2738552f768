"""The C++ host mirror (include/gmmloc_hip/gmm_adapter.hpp: loadGMMModel / trackFrame / optimizeCurrentPose /
associate / renderView+searchCorrespondence / queryPoint, the interface a gmmloc maintainer links) must give what
the Python host gives through the same C-ABI: a g++-built driver runs one frame and the outputs are compared
bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_pose import make_frames
from tests.test_gpu_ba import make_ba_problem, run_gpu
from tests.test_host_cabi import build_adapter_check

pytestmark = pytest.mark.gpu


def test_cpp_adapter_matches_python_host(gpu, map_v1, gt_sync, tmp_path):
    torch, ctx = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    exe = build_adapter_check(tmp_path)
    g0 = api.GMM(ctx, mean, cov)
    g0.save(tmp_path / "m.gmm")
    g = api.GMM.load(ctx, tmp_path / "m.gmm")
    M, N = 700, 400
    f = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 1, M, 900)[0]
    f["octave"][::9] = -1
    uv = np.random.default_rng(1).uniform([0, 0], [752, 480], (N, 2))
    with open(tmp_path / "frame.bin", "wb") as fh:
        np.array([M, N, cam.width, cam.height], np.int32).tofile(fh)
        np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.bf], np.float64).tofile(fh)
        for a in (f["pose_init"], f["Xw"], f["obs"]):
            np.ascontiguousarray(a, np.float64).tofile(fh)
        np.ascontiguousarray(f["octave"], np.int32).tofile(fh)
        uv.tofile(fh)
        # one local window for jointOptimization: 3 free + 2 fixed key-frames, 150 points
        pb = make_ba_problem(mean, cov, gt_sync["V1_01_easy"], cam, 3, 2, 150, 31)
        idx0, d20 = g.associate3d(torch.from_numpy(pb["points"]).cuda(), api.ASSOC_BRUTE)
        pb_assoc = np.where(d20.cpu().numpy() <= 9.0, idx0.cpu().numpy(), -1).astype(np.int32)
        np.array([3, 2, 150, len(pb["obs_pose"])], np.int32).tofile(fh)
        for key, dt in (("poses", np.float64), ("prior", np.uint8), ("points", np.float64)):
            np.ascontiguousarray(pb[key], dt).tofile(fh)
        pb_assoc.tofile(fh)
        for key, dt in (("obs_ptr", np.int32), ("obs_pose", np.int32), ("obs_uvr", np.float64), ("obs_oct", np.int32)):
            np.ascontiguousarray(pb[key], dt).tofile(fh)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(gmmloc_amd._lib.LIB_PATH) + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, str(tmp_path / "m.gmm"), str(tmp_path / "frame.bin"), str(tmp_path / "out.bin")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "components %d" % mean.shape[0] in r.stdout
    print(r.stdout)
    out = open(tmp_path / "out.bin", "rb")
    rd = lambda dt, n: np.fromfile(out, dt, n)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # north-star path
    pose, Xw = T(f["pose_init"][None]), T(f["Xw"][None])
    assoc, _ = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, T(f["obs"][None]), T(f["octave"][None]), want_d2=False)
    torch.cuda.synchronize()
    assert np.array_equal(rd(np.float64, 7), pose.cpu().numpy()[0])
    assert np.array_equal(rd(np.float64, M * 3).reshape(M, 3), Xw.cpu().numpy()[0])
    assert np.array_equal(rd(np.int32, M), assoc.cpu().numpy()[0])
    # ... anchored by the prior edge (the adapter's default)
    pose_a, Xw_a = T(f["pose_init"][None]), T(f["Xw"][None])
    assoc_a, _, _ = gmmloc_amd.track_frames_anchored(ctx, g, cam, prm, pose_a, Xw_a, T(f["obs"][None]), T(f["octave"][None]),
                                                     prior=torch.ones(1, dtype=torch.uint8).cuda(), want_d2=False)
    torch.cuda.synchronize()
    assert np.array_equal(rd(np.float64, 7), pose_a.cpu().numpy()[0]) and not torch.equal(pose_a, pose)
    assert np.array_equal(rd(np.float64, M * 3).reshape(M, 3), Xw_a.cpu().numpy()[0])
    assert np.array_equal(rd(np.int32, M), assoc_a.cpu().numpy()[0])
    # the same host-buffer path from Python (api.HostFramePath: pinned staging, enqueued copies, one synchronize)
    hp = api.HostFramePath(ctx, g, cam, prm, M)
    for _ in range(2):  # the staging buffers are reused between frames
        hpose, hxw = f["pose_init"].copy(), np.ascontiguousarray(f["Xw"]).copy()
        ha = hp.track_frame(hpose, hxw, np.ascontiguousarray(f["obs"]), np.ascontiguousarray(f["octave"], np.int32))
        assert np.array_equal(hpose, pose.cpu().numpy()[0]) and np.array_equal(hxw, Xw.cpu().numpy()[0])
        assert np.array_equal(ha, assoc.cpu().numpy()[0])
    hp.close()
    # optimizeCurrentPose
    pose = T(f["pose_init"][None])
    outl, nin = api.optimize_current_pose(ctx, cam, prm, pose, T(f["Xw"][None]), T(f["obs"][None]), T(f["octave"][None]))
    torch.cuda.synchronize()
    assert np.array_equal(rd(np.float64, 7), pose.cpu().numpy()[0])
    assert rd(np.int32, 1)[0] == int(nin.cpu().numpy()[0])
    assert np.array_equal(rd(np.uint8, M), outl.cpu().numpy()[0])
    # associate
    idx, d2 = g.associate3d(T(f["Xw"]), api.ASSOC_BRUTE)
    assert np.array_equal(rd(np.int32, M), idx.cpu().numpy())
    assert np.array_equal(rd(np.float64, M), d2.cpu().numpy())
    # renderView + searchCorrespondence
    cand, ncand, _, _ = g.search2d(cam, T(f["pose_init"][None]), T(uv[None]), None, k=5)
    assert np.array_equal(rd(np.int32, N), ncand.cpu().numpy()[0])
    assert np.array_equal(rd(np.int32, N * 5).reshape(N, 5), cand.cpu().numpy()[0])
    # queryPoint
    q = g.queryPoint(T(f["Xw"][:1]))
    assert rd(np.int32, 1)[0] == int(q.cpu().numpy().ravel()[0])
    # jointOptimization
    poses, points, dropped, erase, iters = run_gpu(gpu, g, cam, prm, [pb], [pb_assoc])
    nobs = len(pb["obs_pose"])
    assert np.array_equal(rd(np.float64, 5 * 7).reshape(5, 7), poses[0])
    assert np.array_equal(rd(np.float64, 150 * 3).reshape(150, 3), points[0])
    assert np.array_equal(rd(np.uint8, 150), dropped[0])
    assert np.array_equal(rd(np.uint8, nobs), erase[0][:nobs])
    assert rd(np.int32, 1)[0] == int(iters[0])
    assert out.read() == b""
