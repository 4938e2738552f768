"""gl_gmm_create wall time (components, neighbour graph, device-built cell index) for the bench map and the stress map.
    python tools/time_map_build.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
ctx = gmmloc_amd.Context(0)
for K in (4096, 65536):
    mean, cov = synth.synth_gmm(K, 1)
    for rep in range(2):
        t0 = time.perf_counter(); g = api.GMM(ctx, mean, cov); torch.cuda.synchronize(); t1 = time.perf_counter()
        print("K %d gl_gmm_create %.1f ms  index %s" % (K, (t1 - t0) * 1e3, g.index_info()), flush=True)
        g.close()
