#!/bin/bash
# sweep of the tiled W4A16 launch plan (token block x K slices) at a few M
set -e
for mt in 1 2 3 4; do for s in 1 2 4 7 8; do
  QSPEC_TILED_MT=$mt QSPEC_TILED_S=$s timeout -k 10 120 python scripts/bench_tiled.py 64 128 192 512
done; done
