#!/usr/bin/env python3
"""Code-object metadata of every kernel in a built object / the library: registers, spills, scratch, LDS.
    python tools/kernel_meta.py [gmmloc_amd/csrc/gl_ba_fast.o ...]      (default: every .o under gmmloc_amd/csrc)
Unbundles the gfx950 code object from the .hip_fatbin section (objcopy + clang-offload-bundler) and reads the
AMDGPU metadata note (llvm-readelf --notes)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "a.fatbin"), os.path.join(td, "a.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
        if not os.path.getsize(fb):
            return []
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        txt = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    out, cur = [], None
    for line in txt.split("\n"):
        m = re.match(r"\s*(- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        if m.group(1) and cur is not None and "name" in cur:
            out.append(cur)
            cur = None
        if cur is None:
            cur = {}
        cur[m.group(2)] = m.group(3).strip()
    if cur and "name" in cur:
        out.append(cur)
    return [k for k in out if "vgpr_count" in k]


def demangle(n):
    try:
        return subprocess.check_output(["c++filt", n], text=True).strip().replace("(anonymous namespace)::", "").split("(")[0]
    except Exception:
        return n


def main(paths):
    paths = paths or sorted(glob.glob(os.path.join(ROOT, "gmmloc_amd", "csrc", "*.o")))
    print("%-52s %5s %5s %5s %7s %7s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratchB", "ldsB"))
    for p in paths:
        for k in kernels(p):
            print("%-52s %5s %5s %5s %7s %7s %8s %8s" % (demangle(k["name"])[-52:], k["vgpr_count"], k.get("agpr_count", "0"), k["sgpr_count"],
                                                       k.get("vgpr_spill_count", "0"), k.get("sgpr_spill_count", "0"),
                                                       k.get("private_segment_fixed_size", "0"), k.get("group_segment_fixed_size", "0")))


if __name__ == "__main__":
    main(sys.argv[1:])
