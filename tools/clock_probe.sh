#!/bin/bash
# clocks and power under the kernels: rocm-smi sampled beside a bench run; GRBM_GUI_ACTIVE per kernel
set -o pipefail
P=$PWD/gpurun_out/clk
mkdir -p $P
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mix_cost_ubench tools/mix_cost_ubench.hip && timeout -k 10 120 /tmp/mix_cost_ubench > $P/mix_ubench.txt 2>&1 || exit 1
Q="--no-cpu-baseline --no-latency --no-bit128 --no-polymul --no-add32 --no-textbook"
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' '; echo; sleep 0.5; done ) > $P/smi_synthetic.txt 2>&1 &
SMI=$!
timeout -k 10 200 python bench.py --steps 40 $Q > $P/bench_syn.json 2> $P/bench.err || exit 1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $P/smi_refkeys.txt 2>&1 &
SMI=$!
timeout -k 10 200 python bench.py --steps 40 --keys reference $Q > $P/bench_ref.json 2>> $P/bench.err || exit 1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $P/pmc_gui -o vm --output-format csv -- python3 bench.py --steps 2 --warmup 1 $Q > $P/pmc_gui.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
P = "gpurun_out/clk"
cc = glob.glob(P + "/pmc_gui/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob(P + "/pmc_gui/**/*_kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
with open(P + "/gui_active.txt", "w") as f:
    for r in csv.DictReader(open(cc)):
        d = dur.get(r["Dispatch_Id"])
        if d and d > 1e6:
            f.write("%s %s=%s dur_ns=%d ratio=%.4f per ns\n" % (r["Kernel_Name"][:60], r["Counter_Name"], r["Counter_Value"], d, float(r["Counter_Value"]) / d))
PY
rm -rf $P/pmc_gui
echo probe done
