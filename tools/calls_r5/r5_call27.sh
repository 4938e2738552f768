#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py 2>/dev/null | tail -1 | cut -c1-400
