#!/bin/bash
# A/B of route options through the whole pipelined bench: tools/ab_options.sh "tile_split=0" "step_overlap=0" ""
for opt in "$@"; do
  MIA_BENCH_OPTIONS="$opt" python bench.py --no-cpu-baseline --no-secondary --steps 3000 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('options [$opt]', '%.3e' % l['value'], round(l['ms_per_step'],4), 'kernel in loop', round(l['roofline']['kernel_ms'],4), 'alone', round(l['roofline']['kernel_ms_alone'],4), 'serial step', round(l['pipeline']['serial_ms_per_step'],4))"
done
