#!/usr/bin/env python3
"""BASELINE configs[3]: batch replay of all EuRoC V1 / V2 frames (13 735 frame problems, synthetic-from-real-map,
gmmloc_amd/replay.py) through gl_track_frames, frames sharded round-robin over the ranks, results gathered with one
all_gather, TUM trajectories + APE against gt_sync per sequence.
    python tools/replay_euroc.py [--limit N] [--M 300] [--batch 1024] [--out DIR]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/replay_euroc.py ...
One process per GPU; backend nccl (= RCCL over xGMI) when WORLD_SIZE > 1.  Rank 0 prints one JSON line."""
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
    ap.add_argument("--limit", type=int, default=None, help="frames per sequence (default: all)")
    ap.add_argument("--M", type=int, default=300, help="map points per frame")
    ap.add_argument("--batch", type=int, default=1024, help="frames per gl_track_frames call")
    ap.add_argument("--out", default=None, help="directory for the TUM trajectories")
    ap.add_argument("--collective-at-world-1", action="store_true", help="run the RCCL gather / all_reduce even with one rank")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import gmmloc_amd
    from gmmloc_amd import api, replay
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "the replay runs the HIP path: no CPU fallback exists"
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.collective_at_world_1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    cam, prm = api.Camera(), api.Params()
    t0 = time.time()
    maps, frames = replay.materialise_euroc(os.path.join(ROOT, "tests", "golden"), cam, args.M, args.limit)
    t_mat = time.time() - t0
    ctx = gmmloc_amd.Context(local)
    gmms = {name: gmmloc_amd.GMM(ctx, mean, cov, prm) for name, (mean, cov) in maps.items()}
    compute = replay.TrackCompute(ctx, gmms, cam, prm)
    compute(frames[:min(len(frames), 64)])  # warm-up (scratch allocation, kernel load)
    results, dt = replay.replay(frames, compute, rank, world, dist if use_dist else None, torch.device("cuda", local), args.batch,
                                always_collective=args.collective_at_world_1)
    if rank == 0:
        rep = replay.score_euroc(frames, results, args.out)
        print(json.dumps({"config": "configs[3]: batch replay of all V1/V2 frames, round-robin over %d rank(s)" % world,
                          "frames": len(frames), "points_per_frame": args.M, "world": world, "seconds": dt,
                          "frames_per_s": len(frames) / dt, "materialise_s": t_mat, "gathered_bytes_per_frame": 8 * (replay.ROW_D + 1),
                          "backend": "nccl" if use_dist else "none", "sequences": rep}))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
