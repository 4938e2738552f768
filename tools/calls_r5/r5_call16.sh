#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python bench.py --no-cpu-baseline 2> gpurun_out/r5p_bench.err | tail -1 > gpurun_out/r5p_bench.json
python -c "
import json
d=json.loads(open('gpurun_out/r5p_bench.json').read())
r=d['roofline']
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], r['frac'], r['frac_all_points_model'], r['avg_launch_ms'])
print({k:(v['value'] if isinstance(v,dict) and 'value' in v else None) for k,v in d.items() if k.startswith('step_')}, d['roofline'].get('class_1000',{}).get('value'))
print(d['latency'])
"
tail -3 gpurun_out/r5p_bench.err
