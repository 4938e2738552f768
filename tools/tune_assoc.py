"""Sweep the association launch shape (points per thread, K splits) on the GPU."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, gmmloc_amd
    from gmmloc_amd import synth
    from tools.run_configs import ev_time
    N, K = int(sys.argv[2]), int(sys.argv[3])
    ctx = gmmloc_amd.Context(0)
    mean, cov = synth.synth_gmm(K, 1)
    g = gmmloc_amd.GMM(ctx, mean, cov)
    pts = torch.from_numpy(synth.synth_points(mean, cov, N, 2)).cuda()
    t = ev_time(torch, lambda: g.associate3d(pts), 10 if N * K > 1e9 else 200, ctx.stream)
    print(json.dumps({"N": N, "K": K, "chunk": os.environ.get("GMMLOC_ASSOC_CHUNK"), "ppt": os.environ.get("GMMLOC_ASSOC_PPT"), "nsplit": os.environ.get("GMMLOC_ASSOC_NSPLIT"),
                      "us": 1e6 * t, "tflops": 21.0 * N * K / t / 1e12}))
else:
    only = os.environ.get("TUNE_ONLY")  # e.g. TUNE_ONLY=1024000 to sweep a single N
    for N, K in ((1024000, 4096), (50000, 65536), (2000, 4096)):
      if only and str(N) != only:
          continue
      for ch in (8, 16):
        for ppt in (1, 2, 4, 8):
            for ns in (1, 2, 4, 8, 16, 32, 64):
                blocks = -(-N // (256 * ppt)) * ns
                if blocks < 256 or blocks > 20000 or K // ns < 64:
                    continue
                env = dict(os.environ, GMMLOC_ASSOC_PPT=str(ppt), GMMLOC_ASSOC_NSPLIT=str(ns), GMMLOC_ASSOC_CHUNK=str(ch))
                r = subprocess.run([sys.executable, __file__, "child", str(N), str(K)], env=env, capture_output=True, text=True)
                print(r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
