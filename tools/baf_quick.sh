#!/bin/bash
# tools/baf_quick.sh [extra flags]: compile bafd2000 + bafs of gl_ba_fast.hip alone (-DGL_BAF_QUICK, seconds instead of minutes) and print
# their registers / spills - the check to run before a full build when the trial loop changes.  The listing stays in /tmp/baf_quick.s.
cd "$(dirname "$0")/../gmmloc_amd/csrc"
/opt/rocm/bin/hipcc -DGL_BAF_QUICK "$@" -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -disable-machine-sink \
  --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. -S --cuda-device-only gl_ba_fast.hip -o /tmp/baf_quick.s 2>&1 | grep -v hip-link
python3 - <<'PY'
import re
t = open("/tmp/baf_quick.s").read()
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", t):
    print("%-60s vgpr %3s spilled %3s | sgpr %3s spilled %3s | scratch %s B" % (m.group(1)[:60], m.group(5), m.group(6), m.group(3), m.group(4), m.group(2)))
PY
