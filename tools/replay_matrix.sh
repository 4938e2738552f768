#!/bin/bash
# configs[3] replay, anchor x map-noise matrix: APE rmse (mm) of the tracker's pose / of the structure-refined pose per sequence
for A in ${ANCHORS:-none prior fixed}; do for S in ${SIGMAS:-0 0.02}; do
python tools/replay_euroc.py --anchor $A --map-sigma $S ${EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['anchor'], d['map_sigma_m'], 'fps=%.0f'%d['frames_per_s'], ' '.join('%s:%.1f/%.1f'%(k[:5],1e3*v['ape_rmse_m'],1e3*v['ape_rmse_structure_m']) for k,v in d['sequences'].items()))"
done; done
