#!/bin/bash
# The counter pass (rocprofv3 --pmc) dies with the number of OUTSTANDING profiled dispatches (~8 k), not with the
# workload: D/E/F below faulted in round 3 (cycle graphs enqueued back to back), D2/E2 = the same with a host sync per
# cycle.  Kept as the record of how the cause was found; profile_round.sh passes --sync-every-step.
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc32
mkdir -p $out
cd /tmp
run() {
    v=$1; shift
    echo "== $v: $*" >> $out/summary3.txt
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$v -- \
        python3 $root/scripts/profile_cycle.py --plain-engine "$@" > $out/$v.log 2>&1
    echo "rc=$?" >> $out/summary3.txt
    grep -n "profile_cycle\|SIGSEGV\|cycle_ms" $out/$v.log | tail -4 >> $out/summary3.txt
    f=$(find $out/$v -name '*counter_collection.csv' 2>/dev/null | head -1)
    [ -n "$f" ] && echo "$v collected: $(wc -l < $f) counter rows" >> $out/summary3.txt
    rm -rf $out/$v
}
if [ "$1" = "faulting" ]; then
    run D --steps 6 --batch 32 --k 5
    run E --steps 20 --batch 4 --k 3
    run F --steps 12 --batch 4 --k 3
fi
run D2 --steps 6 --batch 32 --k 5 --sync-every-step
run E2 --steps 20 --batch 4 --k 3 --sync-every-step
cat $out/summary3.txt
