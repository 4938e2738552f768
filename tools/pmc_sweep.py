"""Workload for a rocprofv3 --pmc pass over the N x K sweep kernel only (k_assoc_brute): the bench shape
(4096 frames x 2000 points x 4096 Gaussians per launch, as in bench.py) and the stress shape (50 000 x 65 536), 3 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
ctx = gmmloc_amd.Context(0)
for K, N, seed in ((4096, 4096 * 2000, 1), (65536, 50000, 7)):
    mean, cov = synth.synth_gmm(K, seed)
    g = gmmloc_amd.GMM(ctx, mean, cov)
    pts = torch.from_numpy(synth.synth_points(mean, cov, N, seed=3)).cuda()
    for _ in range(3):
        g.associate3d(pts, api.ASSOC_EXHAUSTIVE)
    torch.cuda.synchronize()
print("done")
