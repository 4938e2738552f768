"""Per-basic-block instruction histogram of a gfx950 assembly listing (hipcc -S --cuda-device-only),
used to see what the hot loops of a kernel really issue:  python tools/asm_blocks.py file.s [min_instrs]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 50
blocks, cur = [], None
for ln, l in enumerate(lines):
    l = l.strip()
    m = re.match(r"^(\.LBB\S+|_Z\S+):", l)
    if m:
        cur = [m.group(1)[:24], collections.Counter(), 0, ln + 1]
        blocks.append(cur)
        continue
    if not l or l.startswith(";") or l.startswith(".") or cur is None:
        continue
    cur[1][l.split()[0]] += 1
    cur[2] += 1
for name, c, n, ln in blocks:
    if n < thr:
        continue
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    print("%-24s line %5d n=%4d f64=%4d dpp=%3d ds=%3d glob=%2d scratch=%2d cndmask=%3d mov=%3d lane=%3d salu=%3d cvt=%2d" % (
        name, ln, n, g(lambda k: "f64" in k and "cvt" not in k), g(lambda k: "dpp" in k), g(lambda k: k.startswith("ds_")),
        g(lambda k: k.startswith("global_")), g(lambda k: k.startswith("scratch_")), g(lambda k: "cndmask" in k),
        g(lambda k: k.startswith("v_mov_b32") and "dpp" not in k), g(lambda k: "readfirstlane" in k or "readlane" in k or "writelane" in k),
        g(lambda k: k.startswith("s_")), g(lambda k: "cvt" in k)))
