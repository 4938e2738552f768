"""GPU: the threading contract of the C-ABI (include/gmmloc_hip.h): a gl_gmm_t is immutable and shared, a gl_ctx_t
belongs to one host thread - the reference runs a tracking and a localization thread over one GMM
(gmmloc.cpp:168, localization.cpp:51).  Two host threads with a context each (own stream, own scratch, own driver-query
caches: the dynamic-LDS limit of the 160-KB kernels is set per context) hammer gl_track_frames / gl_joint_optimization /
gl_optimize_current_pose on the same map concurrently; every result must equal the single-threaded one bit for bit."""
import threading

import numpy as np
import pytest

import gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_ba import make_ba_problem, run_gpu
from tests.test_gpu_pose import make_frames

pytestmark = pytest.mark.gpu


def test_two_contexts_two_threads(gpu, map_v1, gt_sync):
    torch, ctx0 = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx0, mean, cov)  # one immutable map, shared by every context
    frames = make_frames(mean, cov, gt_sync["V1_02_medium"], cam, 5, 1500, 4242, outlier_frac=0.05)
    prob = make_ba_problem(mean, cov, gt_sync["V1_01_easy"], cam, 4, 2, 300, 77, True)
    idx, d2 = g.associate3d(torch.from_numpy(prob["points"]).cuda())
    a = np.where(d2.cpu().numpy() <= 9.0, idx.cpu().numpy(), -1).astype(np.int32)

    def work(ctx):
        T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
        out = []
        for rep in range(3):
            pose, Xw = T("pose_init"), T("Xw")
            assoc, dd = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, T("obs"), T("octave"))
            p2 = T("pose_init")
            outl, nin = api.optimize_current_pose(ctx, cam, prm, p2, T("Xw"), T("obs"), T("octave"))
            ba = run_gpu((torch, ctx), g, cam, prm, [prob], [a])
            torch.cuda.synchronize()
            out.append([x.cpu().numpy() for x in (pose, Xw, assoc, dd, p2, outl, nin)] + [ba[0], ba[1], ba[2], ba[3]])
        return out

    ref = work(ctx0)
    for r in ref[1:]:
        for x, y in zip(ref[0], r):
            assert np.array_equal(x, y, equal_nan=True)  # run-to-run deterministic to begin with
    results, errors = {}, []

    def thread_main(name):
        try:
            ctx = gmmloc_amd.Context(0)  # created on the thread that uses it
            results[name] = work(ctx)
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=thread_main, args=(n,)) for n in ("tracking", "localization")]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    for name in ("tracking", "localization"):
        for r in results[name]:
            for x, y in zip(ref[0], r):
                assert np.array_equal(x, y, equal_nan=True), name


def test_two_contexts_oversubscribe_the_latency_shape(gpu, map_v1, gt_sync):
    """Two host threads launch latency-shape refines that EACH want every workgroup slot of the chip (60 frames x 8
    workgroups = 480 of 512) on their own streams, again and again: neither launch can be fully resident while the
    other one is, which is the case a cooperative launch exists for.  The plain launches survive it through the
    rendezvous protocol (frames whose workgroups do not meet within the limit give up untouched, the follow-up kernel
    redoes them): no hang, and every result equals the single-context one bit for bit."""
    torch, ctx0 = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx0, mean, cov)
    frames = make_frames(mean, cov, gt_sync["V1_03_difficult"], cam, 6, 2000, 991, outlier_frac=0.05)
    frames = [frames[i % 6] for i in range(60)]

    def work(ctx, reps):
        T = lambda k: torch.from_numpy(np.stack([f[k] for f in frames])).cuda()
        obs, octv = T("obs"), T("octave")
        out = []
        for rep in range(reps):
            pose, Xw = T("pose_init"), T("Xw")
            assoc, _ = gmmloc_amd.track_frames(ctx, g, cam, prm, pose, Xw, obs, octv, want_d2=False)
            ctx.synchronize()
            out.append([x.cpu().numpy() for x in (pose, Xw, assoc)])
        return out

    ctx0.set_option("ba_shape", 0)
    ref = work(ctx0, 1)[0]
    ctx0.set_option("ba_shape", -1)
    results, errors = {}, []

    def thread_main(name):
        try:
            ctx = gmmloc_amd.Context(0)
            ctx.set_option("ba_shape", 1)
            ctx.set_option("ba_rendezvous_us", 3000)  # keep the test short when the launches do collide
            results[name] = work(ctx, 12)
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=thread_main, args=(n,)) for n in ("a", "b")]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in ts), errors
    for name in ("a", "b"):
        assert len(results[name]) == 12
        for r in results[name]:
            for x, y in zip(ref, r):
                assert np.array_equal(x, y, equal_nan=True), name
