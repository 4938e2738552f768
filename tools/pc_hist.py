"""Histogram of a rocprofv3 PC-sampling CSV: samples per instruction (text), most sampled first, and per opcode class."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print("# %d samples, columns %s" % (len(rows), list(rows[0].keys()) if rows else []))
if not rows:
    sys.exit(0)
ik = next((k for k in rows[0] if k.lower().startswith("instruction") and "comment" not in k.lower()), None)
ck = next((k for k in rows[0] if "comment" in k.lower()), None)
by = collections.Counter()
for r in rows:
    by[(r.get(ik, "?"), r.get(ck, "") if ck else "")] += 1
cls = collections.Counter()
for (ins, _), n in by.items():
    op = ins.split()[0] if ins else "?"
    key = ("f64" if "f64" in op else "s_waitcnt" if op.startswith("s_waitcnt") else "s_barrier" if "barrier" in op else
           "ds" if op.startswith("ds_") else "global" if op.startswith("global_") else "scratch" if op.startswith("scratch_") else
           "salu" if op.startswith("s_") else "valu_other" if op.startswith("v_") else op)
    cls[key] += n
tot = sum(by.values())
print("# by class:", ", ".join("%s %.1f%%" % (k, 100.0 * v / tot) for k, v in cls.most_common()))
for (ins, com), n in by.most_common(150):
    print("%7d %5.2f%%  %-60s %s" % (n, 100.0 * n / tot, ins[:60], com[:80]))
