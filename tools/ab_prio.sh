#!/bin/bash
# A/B of the preparation kernels' wave priority through the whole bench.   gpurun -- 'bash tools/ab_prio.sh'
for lv in 3 0 1; do
echo "== MIA_PREP_PRIO $lv"
for r in 1 2; do
MIA_BUILD_FLAGS=-DMIA_PREP_PRIO=$lv python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('value %.3e ms/step %.4f kernel_ms %.4f alone %.4f serial %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_alone'], j['pipeline']['serial_ms_per_step']))
"
done; done
