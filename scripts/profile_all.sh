#!/bin/bash
# Every BASELINE.json config's shapes at HEAD, one GPU call each group (run ON the GPU box):
#   scripts/profile_all.sh r04 a   -> headline + config 3 (kernel stats, FETCH_SIZE, SQ shares)
#   scripts/profile_all.sh r04 b   -> Llama-2-13B, Llama-3-70B, TinyLlama shapes
tag=${1:-r04}; grp=${2:-a}
run() {  # model batch k
  scripts/profile_round.sh $tag $1 $2 $3 || echo "profile_round $1 $2 $3 exited $?"
  suffix=""; [ "$1 $2 $3" != "llama-3-8b 4 3" ] && suffix="_$1_bs$2_k$3"
  { echo "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ... --kernel-trace -- python3 scripts/profile_cycle.py --steps 6 --model $1 --batch $2 --k $3 --plain-engine --sync-every-step"
    scripts/pmc_cycle_sq.sh $2 $3 $1; } > profiles/${tag}_pmc_sq_cycle${suffix}.txt 2>&1
  cp profiles/${tag}_pmc_sq_cycle${suffix}.txt gpurun_out/prof_$tag/ 2>/dev/null
  echo "== $1 bs=$2 k=$3 done"
}
if [ "$grp" = a ]; then run llama-3-8b 4 3; run llama-3-8b 32 5; else run llama-2-13b 4 3; run llama-3-70b 8 3; run tinyllama-1.1b 1 3; fi
ls gpurun_out/prof_$tag/
