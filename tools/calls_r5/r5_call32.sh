#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/two_trace; rm -rf $O; mkdir -p $O
GMMLOC_BA_TWO_FRAMES=1 rocprofv3 --kernel-trace --stats -d $O -o t -- python tools/refine_only.py 4096 2 0 2000 > $O/out.txt 2> $O/err.txt
DB=$(ls $O/*/*_results.db $O/*_results.db 2>/dev/null | head -1)
python tools/rocpd_summary.py "$DB" | grep -E "k_ba1_fast|kernel" | head
rm -rf $O
