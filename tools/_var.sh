run() { echo "=== $*"; env "$@" timeout -k 10 300 python bench.py --steps 20 --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.3e  ms_per_step %.4f  kernel_ms %.4f  serial %.4f' % (j['value'], j['ms_per_step'], j['roofline'].get('kernel_ms', 0), j['pipeline']['serial_ms_per_step']))
"; }
run A=1
run MIA_NO_STREAM_PICK=1 MIA_PREP_STREAMS=2
run MIA_NO_STREAM_PICK=1 MIA_PREP_STREAMS=3
run MIA_NO_STREAM_PICK=1 MIA_PREP_STREAMS=5
