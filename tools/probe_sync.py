import sys; sys.path.insert(0,'.')
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import synth, api
ctx = gmmloc_amd.Context(0)
for K in (3299, 5096, 4096):
    mean, cov = synth.synth_gmm(K, 1)
    g = api.GMM(ctx, mean, cov)
    for trial in range(3):
        pts = synth.synth_points(mean, cov, 2000, trial)
        p = torch.from_numpy(pts).cuda()
        idx, d2 = g.associate3d(p)
        a = idx.cpu().numpy().copy()
        torch.cuda.synchronize()
        b = idx.cpu().numpy().copy()
        ctx.synchronize()
        c = idx.cpu().numpy().copy()
        print(K, trial, 'immediate==after torch sync', np.array_equal(a, b), 'after ctx sync', np.array_equal(b, c))
        del idx, d2
