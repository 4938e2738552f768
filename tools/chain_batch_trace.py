import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0)
frames = [synth.synth_chain_frame(1200, 1000, 3000, 7000 + b, cam) for b in range(16)]
big = {k: torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(frames[b % 16][k]) for b in range(2048)]).astype(api.CHAIN_DTYPES[k]))).cuda() for k in api.CHAIN_DTYPES}
out = api.track_frame_chain(ctx, cam, prm, big)
for _ in range(4):
    out = api.track_frame_chain(ctx, cam, prm, big, out=out)
    torch.cuda.synchronize()
