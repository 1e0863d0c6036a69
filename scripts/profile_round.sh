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
    --no-cpu-baseline --other-configs none --model $model --batch $batch --k $k > $out/bench.log 2>&1
# (2) decode cycles only (no prompt pass in the process)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cycle -- python3 $root/scripts/profile_cycle.py --steps 20 \
    --model $model --batch $batch --k $k > $out/cycle.log 2>&1
# (3) PMC pass: HBM read bytes per launch.  --sync-every-step: the counter-collection tool faults once ~8 k profiled
# dispatches are outstanding (round 2's SIGSEGV on bs=32 / k=5: six 2.3 k-launch cycle graphs enqueued without a host
# sync; reproduced in round 3 on the headline config with twelve 1.0 k-launch graphs, gone with a sync per cycle).
pmc_ok=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -- python3 $root/scripts/profile_cycle.py --steps 6 \
    --model $model --batch $batch --k $k --plain-engine --sync-every-step > $out/pmc.log 2>&1 || pmc_ok=0
cd $root
suffix=""; [ "$model $batch $k" != "llama-3-8b 4 3" ] && suffix="_${model}_bs${batch}_k${k}"
f=$(find $out/bench -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --other-configs none --model $model --batch $batch --k $k"; grep '^{' $out/bench.log | cut -c1-400; python3 scripts/summarize_prof.py $f; } > profiles/${tag}_bench${suffix}_kernel_stats.txt
f=$(find $out/cycle -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 scripts/profile_cycle.py --steps 20 --model $model --batch $batch --k $k   (decode cycles only)"; cat $out/cycle.log | grep cycle_ms; python3 scripts/summarize_prof.py $f; } > profiles/${tag}_cycle${suffix}_kernel_stats.txt
f=$(find $out/pmc -name '*counter_collection.csv' 2>/dev/null | head -1)
if [ $pmc_ok = 1 ] && [ -n "$f" ]; then
    python3 scripts/pmc_traffic.py $f profiles/${tag}_pmc_fetch_size${suffix}.json $model $batch $k > profiles/${tag}_pmc_fetch_size${suffix}.txt
else   # a dead counter pass must not pass silently: marker under profiles/ and a non-zero exit below
    { echo "PMC pass FAILED for $model bs=$batch k=$k (rocprofv3 --pmc FETCH_SIZE); tail of the log:"; tail -20 $out/pmc.log; } \
        > profiles/${tag}_pmc_fetch_size${suffix}.FAILED.txt
fi
cp profiles/${tag}_*${suffix}* $out/ 2>/dev/null || true
# gpurun merges gpurun_out/ back only below 64 MiB: keep the logs and summaries, drop the raw traces
rm -rf $out/bench $out/cycle $out/pmc
echo "profiles written: $(ls profiles/${tag}_*${suffix}* | tr '\n' ' ')"
[ $pmc_ok = 1 ] || { echo "PMC pass failed (see profiles/${tag}_pmc_fetch_size${suffix}.FAILED.txt)"; exit 3; }
