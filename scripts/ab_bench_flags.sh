for extra in "--no-cpu-baseline" "--no-cpu-baseline --e2e-max-tokens 0" "--no-cpu-baseline --natural-steps 0" "--no-cpu-baseline --e2e-max-tokens 0 --natural-steps 0"; do
python3 bench.py --steps 20 --warmup 5 $extra 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$extra]', d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_shape'])"
done
