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


def test_frame_latency_bounded_next_to_local_ba(gpu, map_v1, gt_sync):
    """The reference runs tracking and the local BA on two threads over one GPU-resident map.  Here a second context keeps
    launching gl_joint_optimization windows back-to-back (8 + 4 key-frames, 1 500 points, the default route: >= 3 000
    observations, i.e. the pipelined shape - five kernels per Levenberg cycle that take the whole chip in turn) while this thread calls the frame-at-a-time path (gl_track_frame_host, latency shape) 300
    times.  A frame whose workgroups cannot meet within ba_rendezvous_us (200 us) falls back to the one-workgroup kernel
    (~0.5 ms): p99 stays below 2 x the idle latency + the rendezvous limit + one BA window + 1 ms (about 4 ms; 2 ms expected) -
    far below the reference's 50 ms frame budget - and every answer is the same bits as on an idle GPU."""
    import time
    torch, ctx0 = gpu
    mean, cov = map_v1
    cam, prm = api.Camera(), api.Params()
    g = api.GMM(ctx0, mean, cov)
    frames = make_frames(mean, cov, gt_sync["V1_03_difficult"], cam, 4, 1200, 606, outlier_frac=0.05)
    prob = make_ba_problem(mean, cov, gt_sync["V1_01_easy"], cam, 8, 4, 1500, 78, True)
    idx, d2 = g.associate3d(torch.from_numpy(prob["points"]).cuda())
    a = np.where(d2.cpu().numpy() <= 9.0, idx.cpu().numpy(), -1).astype(np.int32)
    hp = api.HostFramePath(ctx0, g, cam, prm)

    def one(f):
        pose, X = f["pose_init"].copy(), f["Xw"].copy()
        t0 = time.perf_counter()
        assoc = hp.track_frame(pose, X, f["obs"], f["octave"])
        return time.perf_counter() - t0, (pose, X, assoc)

    ref = [one(f)[1] for f in frames]
    for f in frames:  # warm
        one(f)
    idle = np.array([one(frames[i % 4])[0] for i in range(100)])
    run_gpu((torch, ctx0), g, cam, prm, [prob], [a])  # one BA window alone (warm, then timed): what a frame can queue behind at most
    t0 = time.perf_counter()
    run_gpu((torch, ctx0), g, cam, prm, [prob], [a])
    ba_alone_s = time.perf_counter() - t0
    stop, errors, windows = threading.Event(), [], [0]

    def ba_main():
        try:
            ctx = gmmloc_amd.Context(0)
            while not stop.is_set():
                run_gpu((torch, ctx), g, cam, prm, [prob], [a])
                windows[0] += 1
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    t = threading.Thread(target=ba_main)
    t.start()
    while windows[0] < 2 and t.is_alive():
        time.sleep(0.01)
    ctx0.counter_read(0)
    lat = []
    for i in range(300):
        dt, res = one(frames[i % 4])
        lat.append(dt)
        for x, y in zip(ref[i % 4], res):
            assert np.array_equal(x, y), i
    redone = ctx0.counter_read(0)
    w = windows[0]
    stop.set()
    t.join(timeout=120)
    assert not errors and not t.is_alive(), errors
    lat = np.sort(np.array(lat)) * 1e3
    print("frame latency ms: idle median %.3f | next to %d BA windows: median %.3f p99 %.3f max %.3f, %d of 300 frames redone by the follow-up kernel"
          % (1e3 * np.median(idle), w, np.median(lat), lat[int(0.99 * len(lat))], lat[-1], redone))
    assert w >= 3  # the BA thread really ran alongside
    # the bit-equality above is the hard assertion.  The latency bound is tied to the MECHANISM: a frame that meets the
    # local BA either runs beside it (idle latency) or loses its rendezvous - then it waits out the time limit of the exchange
    # (ba_rendezvous_us) and is redone by the follow-up kernel (another frame's worth) - and in both cases queues behind at most
    # one persistent BA kernel of this window (measured alone above).  A tenfold regression of any of these parts fails here; a busy
    # box (other tenants' processes) only warns below that.
    p99, idle_med = lat[int(0.99 * len(lat))], 1e3 * float(np.median(idle))
    bound = 2.0 * idle_med + 1e-3 * ctx0.get_option("ba_rendezvous_us") + 1e3 * ba_alone_s + 1.0
    assert p99 < bound, (p99, bound, lat[-10:])
    assert redone <= 150  # the fallback is the exception, not the rule (half of the frames at most even next to back-to-back windows)
    if p99 >= 2.0:
        import warnings
        warnings.warn("frame latency p99 %.3f ms next to the local BA (2 ms expected on an idle GPU)" % p99)
