"""Single-problem time of gl_joint_optimization against the workgroups per problem (context option bagen_nb; 0 = the library default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api
from tests.test_gpu_ba import make_ba_problem
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden") + "/map_v1.npz"); mean, cov = d["mean"], d["cov"]
gt = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden") + "/gt_sync.npz")["V1_01_easy"]
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0); g = gmmloc_amd.GMM(ctx, mean, cov, prm)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (P, F, L) in [(1, 1, 200), (2, 2, 400), (4, 2, 800), (8, 4, 1500), (12, 4, 2000), (20, 8, 3000)]:
    p = make_ba_problem(mean, cov, gt, cam, P, F, L, 100)
    idx, d2 = g.associate3d(T(p["points"]))
    assoc = torch.where(d2 <= 9.0, idx, torch.full_like(idx, -1)).reshape(1, L).contiguous()
    args = [T(p["prior"][None]), None, assoc, T(p["obs_ptr"][None]), T(p["obs_pose"][None]), T(p["obs_uvr"][None]), T(p["obs_oct"][None])]
    line = "P%d F%d L%d obs %d:" % (P, F, L, len(p["obs_pose"]))
    for nb in ("0", "1", "2", "4", "8", "16", "32", "64"):
        ctx.set_option("bagen_nb", int(nb))
        def run():
            poses = T(p["poses"][None]); pts = T(p["points"][None])
            return api.joint_optimization(ctx, g, cam, prm, P, F, poses, args[0], pts, assoc, *args[3:])
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
        line += "  NB=%s %.2f" % (nb, (time.perf_counter() - t0) / 5 * 1e3)
    print(line + " ms", flush=True)
