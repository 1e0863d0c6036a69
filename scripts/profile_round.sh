#!/bin/bash
# One GPU call's worth of profiles for a round (run ON the GPU box, from the repo root):
#   scripts/profile_round.sh r02 [model batch k]
# -> gpurun_out/prof_<tag>/...  and the summaries the judge reads, copied to profiles/<tag>_*.
# Counters are collected in their own pass (--pmc with --kernel-trace only), the program directly after `--`.
set -e
tag=${1:-r02}; model=${2:-llama-3-8b}; batch=${3:-4}; k=${4:-3}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out $root/profiles
cd /tmp
# (1) the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 $root/bench.py --steps 50 --warmup 5 \
    --no-cpu-baseline --model $model --batch $batch --k $k > $out/bench.log 2>&1
# (2) decode cycles only (no prompt pass in the process)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cycle -- python3 $root/scripts/profile_cycle.py --steps 20 \
    --model $model --batch $batch --k $k > $out/cycle.log 2>&1
# (3) PMC pass: HBM read bytes per launch
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -- python3 $root/scripts/profile_cycle.py --steps 6 \
    --model $model --batch $batch --k $k --plain-engine > $out/pmc.log 2>&1 || echo "PMC pass failed (see $out/pmc.log)"
cd $root
suffix=""; [ "$model $batch $k" != "llama-3-8b 4 3" ] && suffix="_${model}_bs${batch}_k${k}"
f=$(find $out/bench -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --model $model --batch $batch --k $k"; grep '^{' $out/bench.log | cut -c1-400; python3 scripts/summarize_prof.py $f; } > profiles/${tag}_bench${suffix}_kernel_stats.txt
f=$(find $out/cycle -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 scripts/profile_cycle.py --steps 20 --model $model --batch $batch --k $k   (decode cycles only)"; cat $out/cycle.log | grep cycle_ms; python3 scripts/summarize_prof.py $f; } > profiles/${tag}_cycle${suffix}_kernel_stats.txt
f=$(find $out/pmc -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 scripts/pmc_traffic.py $f profiles/${tag}_pmc_fetch_size${suffix}.json $model $batch $k > profiles/${tag}_pmc_fetch_size${suffix}.txt
cp profiles/${tag}_*${suffix}* $out/ 2>/dev/null || true
# gpurun merges gpurun_out/ back only below 64 MiB: keep the logs and summaries, drop the raw traces
rm -rf $out/bench $out/cycle $out/pmc
echo "profiles written: $(ls profiles/${tag}_*${suffix}* | tr '\n' ' ')"
