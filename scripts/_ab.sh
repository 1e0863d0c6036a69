set -e
export TMPDIR=/tmp
root=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/tl -- python3 $root/scripts/profile_cycle.py --steps 20 --model tinyllama-1.1b --batch 1 --k 3 > $root/gpurun_out/tl.log 2>&1
f=$(find $root/gpurun_out/tl -name '*kernel_stats.csv' | head -1)
python3 $root/scripts/summarize_prof.py $f > $root/gpurun_out/tl.txt
rm -rf $root/gpurun_out/tl
