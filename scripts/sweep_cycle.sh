#!/bin/bash
# Cycle time of the headline model over batch size and context length (decode cycles only, scripts/profile_cycle.py):
# how the cycle moves from a weight stream (small batch, short context) to a KV stream.  Run on the GPU box.
#   scripts/sweep_cycle.sh > gpurun_out/sweep_cycle.txt
for bs in 1 2 4 8 16 32; do for ctx in 128 512 2048 8192; do
  [ $((bs * ctx)) -gt 131072 ] && continue
  line=$(timeout -k 10 300 python scripts/profile_cycle.py --steps 10 --batch $bs --ctx $ctx 2>/dev/null | grep -o "cycle_ms=[0-9.]*")
  echo "llama-3-8b k=3 bs=$bs ctx=$ctx $line"
done; done
