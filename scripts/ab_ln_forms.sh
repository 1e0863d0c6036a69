#!/bin/bash
# A/B of the draft pass's norm forms per batch size (separate norm launch | hand-off prologue | recompute prologue):
#   scripts/ab_ln_forms.sh "8 16"      -> one line per (batch, form): ms per cycle
for bs in $1; do
  for form in "sep:4:1" "handoff:16:1" "recompute:16:0"; do
    IFS=: read name maxm ho <<< "$form"
    QSPEC_FUSE_LN_MAX_M=$maxm QSPEC_LN_HANDOFF=$ho timeout -k 10 300 python3 bench.py --batch $bs --k 3 --steps 20 --warmup 5 \
      --no-cpu-baseline --e2e-max-tokens 0 --natural-steps 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bs=$bs $name', d['ms_per_step'], d['roofline']['per_shape'])"
  done
done
