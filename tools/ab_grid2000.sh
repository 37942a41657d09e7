for setting in "$@"; do
  echo "=== $setting"
  env $setting timeout -k 10 300 python bench.py --steps 2000 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   value %.3e  ms_per_step %.4f  kernel_ms %.4f  steps/launch %.2f' % (j['value'], j['ms_per_step'], j['roofline'].get('kernel_ms', 0), j['roofline'].get('steps_per_launch', 1)))
"
done
