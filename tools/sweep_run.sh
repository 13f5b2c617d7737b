#!/bin/bash
# On the GPU box: times every variant tools/sweep_blind_rotate.py built (twice each).
# usage: tools/sweep_run.sh [--level 80|128] > gpurun_out/sweep.txt
LEVEL=80
if [ "$1" == "--level" ]; then LEVEL=$2; fi
for rep in 1 2; do
for n in $(cat tools/_sweep/names.txt); do
  r=$(TFHE_HIP_LIB=$PWD/tools/_sweep/lib$n.so timeout -k 10 120 python bench.py --level $LEVEL --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-bit128 --no-polymul --no-add32 --no-textbook 2>/dev/null | tail -1 | grep -o "\"value\": [0-9.]*\|\"avg_launch_ms\": [0-9.]*\|\"matches_oracle\": [a-z]*" | tr '\n' ' ')
  echo "$n L$LEVEL $r"
done
done
