#!/bin/bash
# repeat the 70B thread-rank TP check N times with the given env, print pass/fail counts
n=$1; shift
pass=0; fail=0
for i in $(seq $n); do
  if env "$@" timeout -k 10 120 python tests/tp_check.py --family llama-3-70b --threads 8 > /tmp/tpc.log 2>&1; then pass=$((pass+1)); else fail=$((fail+1)); grep "tp vs single" /tmp/tpc.log | grep -v "frac 0.01" | head -3; fi
done
echo "[$*] pass=$pass fail=$fail"
