#!/bin/bash
# On the GPU box: times the isolated product with every variant tools/sweep_variants.py built
# (twice each).  usage: tools/sweep_run_polymul.sh > gpurun_out/sweep_polymul.txt
for rep in 1 2; do
for n in $(cat tools/_sweep/names.txt); do
  r=$(TFHE_HIP_LIB=$PWD/tools/_sweep/lib$n.so timeout -k 10 120 python bench.py --workload polymul 2>/dev/null | tail -1 | grep -o "\"value\": [0-9.]*\|\"frac\": [0-9.]*" | tr '\n' ' ')
  echo "$n $r"
done
done
