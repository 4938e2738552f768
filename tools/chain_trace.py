#!/usr/bin/env python3
"""The kernel timeline of ONE gl_track_frame_chain call (one frame of 1 200 features): which launches a call is made of, how long each runs
and how long the device idles between two dependent launches.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chaintrace -- python tools/chain_trace.py run      (on the GPU box)
    python tools/chain_trace.py table gpurun_out/chaintrace > profiles/r6_chain_trace.txt
`run` makes CALLS calls with a synchronise and a marker launch (a torch fill of a 1-element tensor) between them; `table` takes the LAST
call of the trace."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CALLS = 12


def run():
    import numpy as np
    import torch

    import gmmloc_amd
    from gmmloc_amd import api, synth

    fb = len(sys.argv) > 2 and sys.argv[2] == "fallback"
    cam, prm = api.Camera(), api.Params()
    ctx = gmmloc_amd.Context(0)
    kw = dict(NK=1000, pred_rot_deg=10.0) if fb else {}
    f = synth.synth_chain_frame(1200, 1000, 3000, 7000, cam, **kw)
    from tests.test_gpu_chain import pack as pack_kf
    one = pack_kf(torch, [f])
    out = None
    for _ in range(CALLS):
        out = api.track_frame_chain(ctx, cam, prm, one, out=out)
        torch.cuda.synchronize()
    print("counts", out["counts"].cpu().numpy().tolist())


def table(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + d
    rows = []
    for fn in files:
        with open(fn) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")))
    rows.sort()
    # a call starts with the wrapper's copy of the predicted pose into the kept output buffer (copyBuffer): keep the last call
    starts = [i for i, r in enumerate(rows) if "copyBuffer" in r[2]]
    starts = [i for i in starts if len(rows) - i > 10]
    g = rows[starts[-1] + 1:]
    t0 = g[0][0]
    print("# one gl_track_frame_chain call, one frame of 1 200 features / 1 000 last-frame points / 3 000 local map points: %d launches, %.1f us from the first"
          " kernel's start to the last kernel's end" % (len(g), (g[-1][1] - t0) / 1e3))
    print("# %8s %8s %8s  kernel" % ("start us", "runs us", "idle us"))
    busy, prev = 0, None
    for s, e, n in g:
        n = n.split("(")[0]
        n = n[:110]
        print("  %8.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, n))
        busy += e - s
        prev = e
    print("# kernels run %.1f us, the device idles between them %.1f us" % (busy / 1e3, (g[-1][1] - t0 - busy) / 1e3))
    tot = {}
    for s, e, n in g:
        k = n.split("(")[0].split("<")[0]
        tot[k] = tot.get(k, [0, 0])
        tot[k][0] += 1
        tot[k][1] += e - s
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("#   %-60s x %d  %8.1f us" % (k, c, t / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        table(sys.argv[2])
