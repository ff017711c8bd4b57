#!/bin/bash
cd "$(dirname "$0")/../.."
bad=0
for seed in $(seq $1 $2); do
  for which in fused plane; do for place in end start; do
    out=$(python3 tests/emu/guard_run.py --seed $seed $which $place 2>&1 | tail -n 1); rc=${PIPESTATUS[0]}
    if [ "$out" != "OK" ] && [ "$out" != "SKIP" ]; then bad=$((bad+1)); echo "seed $seed $which $place: $out"; fi
  done; done
  if [ $((seed % 25)) -eq 0 ]; then echo "... seed $seed bad $bad"; fi
done
echo "done $1 $2 bad $bad"
