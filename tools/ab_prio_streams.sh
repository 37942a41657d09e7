#!/bin/bash
# A/B through the whole bench: number of preparation streams (all streams at normal priority).  gpurun -- 'bash tools/ab_prio_streams.sh'
for ps in 3 4 5 6; do
echo "== prep_streams $ps"
for r in 1 2; do
MIA_PREP_STREAMS=$ps python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('value %.3e ms/step %.4f kernel_ms %.4f frac %.3f serial %.4f geo %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['pipeline']['serial_ms_per_step'], j['pipeline']['fixed_geometry']['ms_per_step']))
"
done; done
