#!/usr/bin/env python3
"""Decode the reference's shipped data files into small numeric fixtures.

Run HERE (the container that has /root/reference); the outputs are data only:
  tests/golden/map_v1.npz, map_v2.npz   K x 3 means, K x 9 row-major covariances
                                         (gmmloc_ros/data/map/v{1,2}.gmm)
  tests/golden/gt_sync.npz               per sequence: N x 8 (t x y z qx qy qz qw)
                                         (gmmloc_ros/data/gt_sync/*.txt)
The .gmm framing is the voxblox-style stream written by
gmmloc/src/utils/protobuf_utils.cpp:31-40,82-100: varint32 count, then per
component varint32 size + ComponentProto (gmmloc/proto/gmmloc/GMM.proto:5-14).
This is an independent pure-python reader (the product's reader is C++).
"""
import os, struct, sys
import numpy as np

REF = os.environ.get("GMMLOC_REF", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def read_gmm(path):
    buf = open(path, "rb").read()
    n, pos = varint(buf, 0)
    means, covs = [], []
    for _ in range(n):
        size, pos = varint(buf, pos)
        end = pos + size
        mean, cov = [], []
        while pos < end:
            tag, pos = varint(buf, pos)
            field, wt = tag >> 3, tag & 7
            if wt == 0:
                _, pos = varint(buf, pos)
            elif wt == 2:
                ln, pos = varint(buf, pos)
                vals = struct.unpack("<%dd" % (ln // 8), buf[pos:pos + ln])
                pos += ln
                (mean if field == 3 else cov).extend(vals)
            elif wt == 1:
                (v,) = struct.unpack("<d", buf[pos:pos + 8])
                pos += 8
                (mean if field == 3 else cov).append(v)
            else:
                raise ValueError("wire type %d" % wt)
        assert len(mean) == 3 and len(cov) == 9
        means.append(mean)
        covs.append(cov)
    assert pos == len(buf)
    return np.array(means), np.array(covs)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ("v1", "v2"):
        m, c = read_gmm(os.path.join(REF, "gmmloc_ros/data/map/%s.gmm" % name))
        np.savez_compressed(os.path.join(OUT, "map_%s.npz" % name), mean=m, cov=c)
        print(name, m.shape, c.shape)
    seqs = {}
    d = os.path.join(REF, "gmmloc_ros/data/gt_sync")
    for f in sorted(os.listdir(d)):
        seqs[f[:-4]] = np.loadtxt(os.path.join(d, f))
        print(f, seqs[f[:-4]].shape)
    np.savez_compressed(os.path.join(OUT, "gt_sync.npz"), **seqs)


if __name__ == "__main__":
    main()
