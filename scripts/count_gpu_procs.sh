#!/bin/bash
# Dev tool: while "$@" runs, sample how many processes hold /dev/kfd (the GPU box's process guard allows 6).
"$@" &
pid=$!
max=0
while kill -0 $pid 2>/dev/null; do
  n=0
  for d in /proc/[0-9]*; do
    if ls -l $d/fd 2>/dev/null | grep -q "/dev/kfd"; then n=$((n+1)); fi
  done
  if [ $n -gt $max ]; then max=$n; echo "gpu processes: $n: $(for d in /proc/[0-9]*; do if ls -l $d/fd 2>/dev/null | grep -q /dev/kfd; then tr '\0' ' ' < $d/cmdline | cut -c1-80; echo -n ' | '; fi; done)"; fi
  sleep 0.3
done
wait $pid
echo "exit $? max gpu processes $max"
