#!/bin/bash
# On the GPU box: times every variant tools/sweep_variants.py built (twice each): isolated product,
# Bit80 gates (automatic mode) and, unless --no128, Bit128 gates in mode batch64.
# usage: tools/sweep_run_all.sh [--no128] > gpurun_out/sweep.txt   (results: profiles/r03_sweeps_blind_rotate64.txt)
Q="--no-cpu-baseline --no-latency --no-bit128 --no-polymul --no-add32 --no-textbook"
G='"value": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"matches_oracle": [a-z]*'
for rep in 1 2; do
for n in $(cat tools/_sweep/names.txt); do
  export TFHE_HIP_LIB=$PWD/tools/_sweep/lib$n.so
  r=$(timeout -k 10 120 python bench.py --workload polymul 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | tr '\n' ' ')
  echo "$n polymul $r"
  r=$(timeout -k 10 120 python bench.py --steps 3 --warmup 1 $Q 2>/dev/null | tail -1 | grep -o "$G" | tr '\n' ' ')
  echo "$n L80 $r"
  if [ "$1" != "--no128" ]; then
  r=$(timeout -k 10 120 python bench.py --level 128 --mode batch64 --steps 2 --warmup 1 $Q 2>/dev/null | tail -1 | grep -o "$G" | tr '\n' ' ')
  echo "$n L128/batch64 $r"
  fi
done
done
