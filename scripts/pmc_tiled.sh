#!/bin/bash
# SQ counters of the M-tiled W4A16 GEMM (gemm_w4a16_tiled_kernel) at prefill shapes: two separate --pmc passes
# (8 SQ slots each), kernel trace only.   scripts/pmc_tiled.sh [M ...]      -> gpurun_out/pmc_tiled/*.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_tiled
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
Ms=${@:-2048}
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d $out/p1 -- python3 $root/scripts/bench_tiled.py $Ms > $out/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $out/p1.log; exit 3; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $out/p2 -- python3 $root/scripts/bench_tiled.py $Ms > $out/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $out/p2.log; exit 3; }
cd $root
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f:
        print(p, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "tiled" not in k: continue
        key = (k[:60], r.get("Grid_Size"), r.get("LDS_Block_Size"))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for key, c in agg.items():
        nd = None
        print(p, key)
        for name, v in sorted(c.items()):
            print("    %-28s %.4g" % (name, v))
PY
