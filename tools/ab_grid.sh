#!/bin/bash
# grid over the pipeline's knobs under the round driver's flags: one line per setting (environment of bench.py)
#   gpurun -- 'bash tools/ab_grid.sh "GPU_MAX_HW_QUEUES=4 MIA_ANALYSIS_STREAMS=2 MIA_PREP_STREAMS=1" ...'
for setting in "$@"; do
  echo "=== $setting"
  env $setting timeout -k 10 300 python bench.py --steps 20 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   value %.3e  ms_per_step %.4f  kernel_ms %.4f  steps/launch %.2f  serial %.4f' % (j['value'], j['ms_per_step'], j['roofline'].get('kernel_ms', 0), j['roofline'].get('steps_per_launch', 1), j['pipeline']['serial_ms_per_step']))
"
done
