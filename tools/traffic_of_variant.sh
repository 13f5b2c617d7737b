#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the blind rotation for the given variant libraries (tools/_sweep/lib<name>.so)
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-latency --no-bit128 --no-polymul --no-add32 --no-textbook"
for n in "$@"; do
  export TFHE_HIP_LIB=$PWD/tools/_sweep/lib$n.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$n
    timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$n -o vm --output-format csv -- python3 bench.py --steps 1 --warmup 0 $Q > /tmp/pmc_$n.log 2>&1 || { echo "$n $c failed"; exit 1; }
    python3 - "$n" "$c" <<'PY'
import csv, glob, sys
n, c = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pmc_{n}/**/*_counter_collection.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "blind_rotate" in r["Kernel_Name"] and r["Counter_Name"] == c:
        tot += float(r["Counter_Value"])
mult = 2 if c == "FETCH_SIZE" else 1
print(f"{n} {c} {tot * 1024 * mult / 1e9:.2f} GB per launch (KiB units{', x2 gfx950 correction' if mult == 2 else ''})")
PY
  done
done
