"""Bit fingerprint of gl_joint_optimization over a set of windows: run it with two builds of the library
(GMMLOC_HIP_LIB) to check that a change of the kernel's SCHEDULE left every output bit where it was."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_ba import make_ba_problem
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
d = np.load(G + "/map_v1.npz"); mean, cov = d["mean"], d["cov"]
gt = np.load(G + "/gt_sync.npz")["V1_01_easy"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
if os.environ.get("BAGEN_MODE"):
    ctx.set_option("bagen_mode", int(os.environ["BAGEN_MODE"]))  # 1 persistent kernel, 2 pipelined shape
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (P, F, L, prior) in [(1, 0, 120, False), (1, 1, 200, True), (2, 2, 400, True), (4, 2, 800, False), (8, 4, 1500, True),
                         (12, 4, 2000, True), (20, 8, 3000, True), (21, 2, 1000, True), (22, 2, 1000, True)]:
    for seed in (100, 101, 102):
        p = make_ba_problem(mean, cov, gt, cam, P, F, L, seed, prior)
        idx, d2 = g.associate3d(T(p["points"]))
        assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).contiguous()
        for nb in (0, 1):
            ctx.set_option("bagen_nb", nb)
            poses, pts = T(p["poses"][None]), T(p["points"][None])
            out = api.joint_optimization(ctx, g, cam, prm, P, F, poses, T(p["prior"][None]), pts, assoc, T(p["obs_ptr"][None]),
                                         T(p["obs_pose"][None]), T(p["obs_uvr"][None]), T(p["obs_oct"][None]))
            torch.cuda.synchronize()
            h = hashlib.sha1(poses.cpu().numpy().tobytes() + pts.cpu().numpy().tobytes()
                             + b"".join(x.cpu().numpy().tobytes() for x in out if hasattr(x, "cpu"))).hexdigest()[:16]
            print("P%d F%d L%d seed %d nb %d  %s" % (P, F, L, seed, nb, h), flush=True)
