#!/bin/bash
export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np, torch, gmmloc_amd
from gmmloc_amd import api, synth
mean, cov = synth.synth_gmm(4096, 43)
ctx = gmmloc_amd.Context(0)
for c8 in (1,0):
    ctx.set_option("assoc_cell8", c8)
    print("opt", ctx.get_option("assoc_cell8"), ctx.get_option("assoc_pack_mb"))
    g = api.GMM(ctx, mean, cov)
    print(c8, g.index_info())
PY
