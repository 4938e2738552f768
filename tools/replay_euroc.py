#!/usr/bin/env python3
"""BASELINE configs[3]: batch replay of all EuRoC V1 / V2 frames (13 735 frame problems, synthetic-from-real-map,
gmmloc_amd/replay.py) through gl_track_frames, frames sharded round-robin over the ranks, results gathered with one
all_gather, TUM trajectories + APE against gt_sync per sequence.
    python tools/replay_euroc.py [--gpus N] [--limit N] [--M 300] [--batch 1024] [--out DIR]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/replay_euroc.py --gpus 8 ...
One process per GPU (--gpus N starts the ranks itself when no launcher did, gmmloc_amd/launch.py); backend nccl
(= RCCL over xGMI) when there is more than one rank.  Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 without a launcher: the ranks are started here")
    ap.add_argument("--limit", type=int, default=None, help="frames per sequence (default: all)")
    ap.add_argument("--M", type=int, default=300, help="map points per frame")
    ap.add_argument("--batch", type=int, default=1024, help="frames per gl_track_frames call")
    ap.add_argument("--out", default=None, help="directory for the TUM trajectories")
    ap.add_argument("--map-sigma", type=float, default=0.0, help="noise (m) on the map points the tracker is given (0: exact points)")
    ap.add_argument("--anchor", default="prior", choices=["none", "prior", "fixed"], help="gauge anchor of the structure refine")
    ap.add_argument("--fixed", type=int, default=2, help="fixed observer key-frames per frame (--anchor fixed)")
    ap.add_argument("--collective-at-world-1", action="store_true", help="run the RCCL gather / all_reduce even with one rank")
    args = ap.parse_args()
    from gmmloc_amd import launch
    if args.gpus > 1 and not launch.is_rank():
        sys.exit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    import torch
    import gmmloc_amd
    from gmmloc_amd import api, replay
    ranks = launch.Ranks("nccl").init(always=args.collective_at_world_1)
    world, rank, local, dist = ranks.world, ranks.rank, ranks.local, ranks.dist
    cam, prm = api.Camera(), api.Params()
    t0 = time.time()
    maps, frames = replay.materialise_euroc(os.path.join(ROOT, "tests", "golden"), cam, args.M, args.limit, map_sigma=args.map_sigma,
                                            fixed=args.fixed if args.anchor == "fixed" else 0)
    t_mat = time.time() - t0
    ctx = gmmloc_amd.Context(local)
    gmms = {name: gmmloc_amd.GMM(ctx, mean, cov, prm) for name, (mean, cov) in maps.items()}
    compute = replay.TrackCompute(ctx, gmms, cam, prm, anchor=args.anchor)
    compute(frames[:min(len(frames), 64)])  # warm-up (scratch allocation, kernel load)
    results, dt = replay.replay(frames, compute, rank, world, dist, ranks.device, args.batch,
                                always_collective=args.collective_at_world_1)
    if rank == 0:
        rep = replay.score_euroc(frames, results, args.out)
        print(json.dumps({"config": "configs[3]: batch replay of all V1/V2 frames, round-robin over %d rank(s)" % world,
                          "frames": len(frames), "points_per_frame": args.M, "world": world, "seconds": dt,
                          "frames_per_s": len(frames) / dt, "materialise_s": t_mat, "gathered_bytes_per_frame": 8 * (replay.ROW_D + 1),
                          "backend": "nccl" if dist is not None else "none", "map_sigma_m": args.map_sigma, "anchor": args.anchor,
                          "sequences": rep}))
    ranks.close()


if __name__ == "__main__":
    main()
