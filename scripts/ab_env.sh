#!/bin/bash
# A/B of runtime environment knobs on the default bench workload: scripts/ab_env.sh "A=1" "B=0 C=1" ...
for e in "" "$@"; do
  env $e timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --e2e-max-tokens 0 --natural-steps 0 ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$e]', d['ms_per_step'], d['roofline']['per_shape'])"
done
