#!/bin/bash
# A/B of build variants through the whole pipelined bench (value, ms/step, kernel ms in the loop, serial ms/step)
for flags in "$@"; do
  MIA_BUILD_FLAGS="$flags" python bench.py --no-cpu-baseline --no-secondary --steps 3000 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('flags [$flags]', '%.3e' % l['value'], round(l['ms_per_step'],4), round(l['roofline']['kernel_ms'],4), round(l['roofline']['kernel_ms_alone'],4), round(l['pipeline']['serial_ms_per_step'],4))"
done
