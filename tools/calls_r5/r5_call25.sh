#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
BAGEN_MODE=1 timeout 300 python tools/ba_bits.py 2>&1 | grep -v amdgpu.ids | head -12
