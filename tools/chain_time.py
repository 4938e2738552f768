#!/usr/bin/env python3
"""One tracked frame: gl_track_frame_chain (device resident) against the same four stages as four host calls with the glue on the
host's side of the device (torch ops) and a synchronise between the stages - what a host that consumes each stage's result does:
    python tools/chain_time.py [NF NL NP]
Prints the latency of one frame both ways, the single-call latencies of the four stages, and the batch throughput of the chain."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gmmloc_amd
from gmmloc_amd import api, synth

NF, NL, NP = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (1200, 1000, 3000)
cam, prm = api.Camera(), api.Params()
ctx = gmmloc_amd.Context(0)


def pack(frames):
    return {k: torch.from_numpy(np.ascontiguousarray(np.stack([f[k] for f in frames]).astype(api.CHAIN_DTYPES[k]))).cuda() for k in api.CHAIN_DTYPES}


def median_ms(fn, n=30, skip=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts[skip:]))


def four_calls(a, sync=True):
    """the same work as four entry points; the glue as torch ops on the device, a synchronise after every stage"""
    B = a["feat_oct"].shape[0]
    s = (lambda: torch.cuda.synchronize()) if sync else (lambda: None)
    with torch.cuda.stream(ctx.stream):
        m1, n1 = api.search_by_projection_frame(ctx, cam, a["pose_cw"], a["pose_lw"], a["feat_uv"], a["feat_ur"], a["feat_oct"], a["feat_angle"], a["feat_desc"],
                                                a["feat_taken"], a["last_pt"], a["last_valid"], a["last_oct"], a["last_angle"], a["last_desc"], th=7.0)
        s()
        has = m1 >= 0
        idx = m1.clamp(min=0).long()
        Xw = torch.where(has[..., None], torch.gather(a["last_pt"], 1, idx[..., None].expand(-1, -1, 3)), torch.zeros((), dtype=torch.float64, device="cuda"))
        obs = torch.cat([a["feat_uv"], a["feat_ur"][..., None].double()], 2).contiguous()
        oc = torch.where(has, a["feat_oct"], torch.full_like(a["feat_oct"], -1))
        pose = a["pose_cw"].clone()
        outl = torch.zeros_like(a["feat_taken"])
        gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose, Xw.contiguous(), obs, oc.contiguous(), outl)
        s()
        kept = torch.where(outl != 0, torch.full_like(m1, -1), m1)
        taken = ((a["feat_taken"] != 0) | (kept >= 0)).to(torch.uint8)
        l = torch.gather(a["last_to_local"], 1, idx)
        cand = a["mp_cand"].clone()
        cand.scatter_(1, torch.where(has & (l >= 0), l, torch.full_like(l, NP)).long().clamp(max=NP - 1), torch.zeros_like(taken))  # (approximate glue: timing only)
        q, t = pose[:, :4], pose[:, 4:]
        qv = -q[:, :3]
        uv = 2 * torch.cross(qv, -t, dim=1)
        twc = (-t + q[:, 3:4] * uv + torch.cross(qv, uv, dim=1)).contiguous()
        m3 = api.search_local_points(ctx, cam, a["feat_uv"], a["feat_ur"], a["feat_oct"], a["feat_desc"], taken, pose, twc, a["mp_pos"], a["mp_normal"],
                                     a["mp_max_dist"], a["mp_min_dist"], cand, a["mp_desc"], th=3.0)[0]
        s()
        hm = m3 >= 0
        Xw2 = torch.where((kept >= 0)[..., None], Xw, torch.where(hm[..., None], torch.gather(a["mp_pos"], 1, m3.clamp(min=0).long()[..., None].expand(-1, -1, 3)),
                                                                 torch.zeros((), dtype=torch.float64, device="cuda")))
        oc2 = torch.where((kept >= 0) | hm, a["feat_oct"], torch.full_like(a["feat_oct"], -1))
        gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose, Xw2.contiguous(), obs, oc2.contiguous(), outl)
        s()
    return pose


frames = [synth.synth_chain_frame(NF, NL, NP, 7000 + b, cam) for b in range(64)]
one = pack(frames[:1])
chain1 = lambda: api.track_frame_chain(ctx, cam, prm, one)
with torch.cuda.stream(ctx.stream):
    t_chain = median_ms(chain1)
    # the latency of a frame follows the number of Levenberg trials its two pose optimisations take (the tail of a converged optimisation
    # is decided by rounding: 37 .. 69 evaluations on like frames): the mean over 16 frames beside the first one's
    t_each = []
    for b in range(16):
        fb = pack(frames[b:b + 1])
        t_each.append(median_ms(lambda: api.track_frame_chain(ctx, cam, prm, fb), n=12, skip=2))
t_four = median_ms(lambda: four_calls(one))
t_four_nosync = median_ms(lambda: four_calls(one, sync=False))
# the stages alone (one call + synchronise each), on the chain's own intermediate shapes
with torch.cuda.stream(ctx.stream):
    t_s1 = median_ms(lambda: api.search_by_projection_frame(ctx, cam, one["pose_cw"], one["pose_lw"], one["feat_uv"], one["feat_ur"], one["feat_oct"], one["feat_angle"],
                                                             one["feat_desc"], one["feat_taken"], one["last_pt"], one["last_valid"], one["last_oct"], one["last_angle"],
                                                             one["last_desc"], th=7.0))
    out = chain1()
    torch.cuda.synchronize()
    has = out["match_last"] >= 0
    Xw = torch.where(has[..., None], torch.gather(one["last_pt"], 1, out["match_last"].clamp(min=0).long()[..., None].expand(-1, -1, 3)),
                     torch.zeros((), dtype=torch.float64, device="cuda")).contiguous()
    obs = torch.cat([one["feat_uv"], one["feat_ur"][..., None].double()], 2).contiguous()
    oc = torch.where(has, one["feat_oct"], torch.full_like(one["feat_oct"], -1)).contiguous()
    outl = torch.zeros_like(one["feat_taken"])
    pose = one["pose_cw"].clone()
    t_s2 = median_ms(lambda: gmmloc_amd.optimize_current_pose(ctx, cam, prm, pose.copy_(one["pose_cw"]), Xw, obs, oc, outl))
    twc = torch.zeros((1, 3), dtype=torch.float64, device="cuda")
    t_s3 = median_ms(lambda: api.search_local_points(ctx, cam, one["feat_uv"], one["feat_ur"], one["feat_oct"], one["feat_desc"], one["feat_taken"], out["pose_mm"], twc,
                                                     one["mp_pos"], one["mp_normal"], one["mp_max_dist"], one["mp_min_dist"], one["mp_cand"], one["mp_desc"], th=3.0))
# round 6: the same frame with the reference key-frame's buffers attached (the fallback's launches return at once: what the option
# costs a frame that tracks), and a frame that NEEDS the fallback (prediction 10 degrees off: trackKeyFrame, tracking.cpp:297-331)
from tests.test_gpu_chain import pack as pack_kf  # noqa: E402
kf_ok = pack_kf(torch, [synth.synth_chain_frame(NF, NL, NP, 7000, cam, NK=NL)])
kf_fb = pack_kf(torch, [synth.synth_chain_frame(NF, NL, NP, 7000, cam, NK=NL, pred_rot_deg=10.0)])
with torch.cuda.stream(ctx.stream):
    t_kf_ok = median_ms(lambda: api.track_frame_chain(ctx, cam, prm, kf_ok))
    t_kf_fb = median_ms(lambda: api.track_frame_chain(ctx, cam, prm, kf_fb))
    o_fb = api.track_frame_chain(ctx, cam, prm, kf_fb)
    torch.cuda.synchronize()
    t_front = median_ms(lambda: api.track_frame_chain_front(ctx, cam, prm, kf_ok))
    fr_out = api.track_frame_chain_front(ctx, cam, prm, kf_ok)
    torch.cuda.synchronize()
    t_back = median_ms(lambda: api.track_frame_chain_back(ctx, cam, prm, kf_ok, {k: v.clone() for k, v in fr_out.items()}))
# round 6: gl_optimize_current_pose compacts the frame's 1 200-slot problems itself (option pose_compact); the same with the option off
t_nc = []
ctx.set_option("pose_compact", 0)
with torch.cuda.stream(ctx.stream):
    for b in range(16):
        fb = pack(frames[b:b + 1])
        t_nc.append(median_ms(lambda: api.track_frame_chain(ctx, cam, prm, fb), n=12, skip=2))
ctx.set_option("pose_compact", -1)
# ... and with the caller's OUTPUT buffers kept from frame to frame (`out=`: no allocations in the wrapper)
t_one = []
with torch.cuda.stream(ctx.stream):
    for b in range(16):
        fb = pack(frames[b:b + 1])
        keep = api.track_frame_chain(ctx, cam, prm, fb)
        t_one.append(median_ms(lambda: api.track_frame_chain(ctx, cam, prm, fb, out=keep), n=14, skip=4))
t_keep = [float(np.mean(t_one)), float(np.min(t_one)), float(np.max(t_one))]
B = 2048
big = pack([frames[b % 64] for b in range(B)])
ctx.set_option("pose_compact", 0)
with torch.cuda.stream(ctx.stream):
    t_batch_nc = median_ms(lambda: api.track_frame_chain(ctx, cam, prm, big), n=8, skip=2)
ctx.set_option("pose_compact", -1)
with torch.cuda.stream(ctx.stream):
    t_batch = median_ms(lambda: api.track_frame_chain(ctx, cam, prm, big), n=8, skip=2)
print(json.dumps({"config": "one tracked frame (trackWithMotionModel -> searchLocalPoints -> trackLocalMap): %d features, %d last-frame map points, %d local map points" % (NF, NL, NP),
                  "chain_one_frame_ms": t_chain, "chain_one_frame_ms_mean_of_16_frames": float(np.mean(t_each)), "chain_one_frame_ms_min_max_of_16": [float(np.min(t_each)), float(np.max(t_each))],
                  "four_calls_with_sync_between_ms": t_four, "four_calls_enqueued_without_sync_ms": t_four_nosync,
                  "single_call_ms": {"searchByProjection(frame)": t_s1, "optimizeCurrentPose": t_s2, "searchLocalPoints": t_s3, "sum_of_four": t_s1 + 2 * t_s2 + t_s3},
                  "chain_batch_frames_per_s": B / (t_batch * 1e-3), "batch": B,
                  "with_key_frame_buffers_frame_that_tracks_ms": t_kf_ok, "frame_through_trackKeyFrame_fallback_ms": t_kf_fb,
                  "fallback_frame_mode_and_counts2": [int(v) for v in o_fb["counts2"][0].cpu().numpy()],
                  "two_halves_ms": {"front": t_front, "back": t_back},
                  "pose_compact_off_one_frame_ms_mean_min_max_of_16": [float(np.mean(t_nc)), float(np.min(t_nc)), float(np.max(t_nc))],
                  "pose_compact_off_batch_frames_per_s": B / (t_batch_nc * 1e-3),
                  "buffers_kept_one_frame_ms_mean_min_max_of_16": t_keep,
                  "note": "wall clock incl. the Python wrapper, median of 25; the four-call form does its glue as torch ops on the device"}))
