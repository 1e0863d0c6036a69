#!/bin/bash
# SQ counters of the headline cycle's kernels (one --pmc pass, kernel trace only, a host sync per cycle: see DESIGN.md on
# the counter tool's limit of outstanding dispatches).   scripts/pmc_cycle_sq.sh [batch k]  -> stdout
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_cycle_sq
batch=${1:-4}; k=${2:-3}; model=${3:-llama-3-8b}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d $out/p -- python3 $root/scripts/profile_cycle.py --steps 6 --model $model --batch $batch --k $k --plain-engine --sync-every-step > $out/p.log 2>&1 || { echo "pass failed"; tail -5 $out/p.log; exit 3; }
cd $root
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/p/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "")[:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print("%-66s %7s %12s %8s %8s %8s %8s %8s" % ("kernel", "launches", "wave_qcyc/l", "wait%", "istall%", "active%", "vmem%", "lds%"))
for k, c in rows[:24]:
    w = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-66s %7d %12.0f %8.1f %8.1f %8.1f %8.1f %8.1f" % (k, len(n[k]), w / len(n[k]), 100 * c.get("SQ_WAIT_ANY", 0) / w,
          100 * c.get("SQ_WAIT_INST_ANY", 0) / w, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w, 100 * c.get("SQ_ACTIVE_INST_VMEM", 0) / w,
          100 * c.get("SQ_ACTIVE_INST_LDS", 0) / w))
PY
rm -rf $out/p
