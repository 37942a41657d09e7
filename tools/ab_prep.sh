for ps in 2 3 4; do for d in 6 8; do
echo "== prep_streams $ps depth $d"
MIA_PREP_STREAMS=$ps python bench.py --no-cpu-baseline --no-secondary --pipeline-depth $d 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('value %.3e ms/step %.4f kernel_ms %.4f alone %.4f serial %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_alone'], j['pipeline']['serial_ms_per_step']))
"
done; done
