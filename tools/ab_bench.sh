#!/bin/bash
# A/B through the whole bench under the round driver's flags: one line per environment setting.
#   gpurun -- 'bash tools/ab_bench.sh "MIA_PREP_STREAMS=2" "MIA_PREP_STREAMS=5" "MIA_PIPELINE_DEPTH=4"'
# (profiles/r04_stream_ab.txt was made with it: 2 / 3 / 5 preparation streams, with and without round 3's queue probing)
for setting in "A=1" "$@"; do
  echo "=== $setting"
  env $setting timeout -k 10 300 python bench.py --steps 20 --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.3e  ms_per_step %.4f  kernel_ms %.4f  serial %.4f' % (j['value'], j['ms_per_step'], j['roofline'].get('kernel_ms', 0), j['pipeline']['serial_ms_per_step']))
"
done
