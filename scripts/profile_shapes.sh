#!/bin/bash
# Kernel-trace summaries of the decode cycle for several (model, batch, k) workloads in ONE GPU call (run ON the GPU box):
#   scripts/profile_shapes.sh r04a "llama-2-13b 4 3" "llama-3-70b 8 3" "tinyllama-1.1b 1 3"
# -> profiles/<tag>_cycle_<model>_bs<b>_k<k>_kernel_stats.txt  (decode cycles only; no PMC pass: see profile_round.sh)
set -e
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p $root/profiles
for wl in "$@"; do
  set -- $wl; model=$1; batch=$2; k=$3
  out=$root/gpurun_out/prof_${tag}_${model}_bs${batch}_k${k}
  mkdir -p $out
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/cycle -- python3 $root/scripts/profile_cycle.py \
      --steps 10 --model $model --batch $batch --k $k > $out/cycle.log 2>&1 )
  f=$(find $out/cycle -name '*kernel_stats.csv' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 scripts/profile_cycle.py --steps 10 --model $model --batch $batch --k $k   (decode cycles only)"
    grep cycle_ms $out/cycle.log; python3 scripts/summarize_prof.py $f; } > profiles/${tag}_cycle_${model}_bs${batch}_k${k}_kernel_stats.txt
  cp profiles/${tag}_cycle_${model}_bs${batch}_k${k}_kernel_stats.txt $out/
  rm -rf $out/cycle
  echo "done $wl: $(grep cycle_ms $out/cycle.log)"
done
