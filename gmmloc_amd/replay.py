"""Offline batch replay sharded over the GPUs of one node (BASELINE configs[3]).

Frame problems are independent units: rank r owns frames r, r+P, r+2P, ... (round-robin), the
GMM is replicated, and there is NO data-path collective.  After the local work each rank
contributes its per-frame results (pose 7 doubles + 2 counters) to one all_gather and the
timing to one all_reduce(MAX).  Works with backend "nccl" (= RCCL over xGMI) on GPUs and with
"gloo" on CPU (the latter is what the CPU tests exercise: sharding, padding of uneven shards,
gather order -- the compute callback is supplied by the caller).
"""
import numpy as np


def shard_indices(n_frames, rank, world):
    """Round-robin frame -> rank map: frame i goes to rank i % world."""
    return np.arange(rank, n_frames, world, dtype=np.int64)


def gather_results(local_idx, local_vals, n_frames, world, dist=None, device="cpu"):
    """All-gather per-frame rows (local_vals: (len(local_idx), D) float64) into a (n_frames, D) array
    in frame order on every rank.  Shards are padded to the common maximum length."""
    import torch
    D = local_vals.shape[1]
    if world == 1 or dist is None:
        out = np.zeros((n_frames, D))
        out[local_idx] = local_vals
        return out
    cap = (n_frames + world - 1) // world
    buf = torch.zeros((cap, D + 1), dtype=torch.float64, device=device)
    buf[:, 0] = -1.0
    if len(local_idx):
        buf[:len(local_idx), 0] = torch.as_tensor(local_idx, dtype=torch.float64, device=device)
        buf[:len(local_idx), 1:] = torch.as_tensor(local_vals, dtype=torch.float64, device=device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    allp = torch.cat(parts).cpu().numpy()
    allp = allp[allp[:, 0] >= 0]
    out = np.zeros((n_frames, D))
    out[allp[:, 0].astype(np.int64)] = allp[:, 1:]
    return out


def replay(frames, compute, rank=0, world=1, dist=None, device="cpu", batch=256):
    """frames: list of frame problems; compute(list_of_frames) -> (len, D) float64 rows.
    Returns (results (n_frames, D) on every rank, elapsed seconds = max over ranks)."""
    import time
    import torch
    idx = shard_indices(len(frames), rank, world)
    rows = []
    if dist is not None and world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for s in range(0, len(idx), batch):
        rows.append(compute([frames[i] for i in idx[s:s + batch]]))
    dt = time.perf_counter() - t0
    local = np.concatenate(rows) if rows else np.zeros((0, 1))
    if local.shape[0] == 0:  # a rank may own no frames
        D = 1
        probe = torch.tensor([0], dtype=torch.int64, device=device)
        local = np.zeros((0, D))
    t = torch.tensor([dt, float(local.shape[1])], dtype=torch.float64, device=device)
    if dist is not None and world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    D = int(t[1].item())
    if local.shape[0] == 0:
        local = np.zeros((0, D))
    return gather_results(idx, local, len(frames), world, dist, device), float(t[0].item())


# ---- K-sharded association (the stress shape: 50 000 points x 65 536 Gaussians) --------------
# Rank r holds the Gaussians [k_offset, k_offset + K_r) and evaluates the exact chi2 argmin of every point over
# its shard (gl_associate3d: the per-pair chi2 does not depend on which other Gaussians are present, so the
# values are bit-identical to the unsharded sweep).  The exchange step is the argmin merge: one all_reduce(MIN)
# on the N fp64 chi2 values, then one all_reduce(MIN) on the N global indices of the ranks that hold the
# minimum -- the lowest index wins ties, exactly like the sequential sweep.  (fp64 chi2 + index do not fit the
# packed 64-bit key of the fp32 variant sketched in SURVEY 8e, hence two collectives of N x 8 bytes.)
INDEX_NONE = np.iinfo(np.int64).max


def shard_components(K, rank, world):
    """Contiguous K-shard of rank: (k_offset, K_r); the first K % world ranks hold one more."""
    base, extra = divmod(K, world)
    k0 = rank * base + min(rank, extra)
    return k0, base + (1 if rank < extra else 0)


def merge_sharded_association(d2_local, idx_local, k_offset, dist=None, world=1):
    """d2_local (N,) float64 and idx_local (N,) int (index inside the shard, < 0 = none) as torch tensors on the
    collective's device.  Returns (global idx int64 with -1 = none, d2) on every rank."""
    import torch
    d2 = d2_local.clone()
    valid = idx_local >= 0
    d2 = torch.where(valid, d2, torch.full_like(d2, float("inf")))
    mine = d2.clone()
    if dist is not None and world > 1:
        dist.all_reduce(d2, op=dist.ReduceOp.MIN)
    gi = torch.where(valid & (mine == d2), idx_local.to(torch.int64) + int(k_offset),
                     torch.full_like(idx_local, INDEX_NONE, dtype=torch.int64))
    if dist is not None and world > 1:
        dist.all_reduce(gi, op=dist.ReduceOp.MIN)
    gi = torch.where(gi == INDEX_NONE, torch.full_like(gi, -1), gi)
    return gi, d2
