#!/bin/bash
# A/B through the whole bench: analysis streams x preparation streams x depth.   gpurun -- 'bash tools/ab_streams.sh'
for as in 1 2 3; do for ps in 3 4; do for d in 8 12; do
echo "== analysis_streams $as prep_streams $ps depth $d"
MIA_ANALYSIS_STREAMS=$as MIA_PREP_STREAMS=$ps python bench.py --no-cpu-baseline --no-secondary --pipeline-depth $d 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('value %.3e ms/step %.4f kernel_ms %.4f alone %.4f serial %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_alone'], j['pipeline']['serial_ms_per_step']))
"
done; done; done
