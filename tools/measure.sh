#!/bin/bash
# One measurement pass on the GPU box: bench lines, rocprofv3 kernel statistics and the PMC
# passes (each counter group in a run of its own, no trace domains beside --kernel-trace).
# usage: tools/measure.sh <pass-name>     results under gpurun_out/<pass-name>/
# tools/collect_profiles.py copies the summaries into profiles/.
set -o pipefail
P=$PWD/gpurun_out/$1
mkdir -p $P
export TMPDIR=/tmp
T="timeout -k 10"
Q="--no-cpu-baseline --no-latency --no-bit128 --no-polymul --no-add32 --no-textbook"
$T 400 python bench.py --steps 10 > $P/bench80.json 2> $P/bench80.err || exit 1
$T 200 python bench.py --level 128 --steps 5 --no-polymul --no-add32 --no-textbook > $P/bench128.json 2>> $P/bench80.err || exit 1
$T 200 python bench.py --keys reference --steps 5 $Q > $P/bench80_refkeys.json 2>> $P/bench80.err || exit 1
$T 300 python bench.py --workload add32 --steps 1 --warmup 0 > $P/add32.json 2>> $P/bench80.err || exit 1
$T 200 python bench.py --workload polymul > $P/polymul.json 2>> $P/bench80.err || exit 1
# BASELINE config 4 on one GPU: 2^20 gates per step in launches of 16 384
$T 300 python bench.py --total-gates 1048576 --steps 1 --warmup 0 $Q > $P/strong_1M_gates.json 2>> $P/bench80.err || exit 1
# whole-gate time by batch size and mapping (thresholds of the automatic mode)
$T 300 python tools/time_batch_sizes.py 1 64 256 257 512 513 768 769 1024 1025 1280 1536 1537 2048 2049 2304 2816 3072 3300 4096 5000 > $P/batch_sizes.txt 2>> $P/bench80.err || exit 1
$T 300 python tools/time_batch_sizes.py --level 128 1 256 512 768 769 1024 1025 1536 1537 2048 2304 3072 3300 4096 >> $P/batch_sizes.txt 2>> $P/bench80.err || exit 1
$T 300 rocprofv3 --kernel-trace --stats -d $P/prof -o vm --output-format csv -- python3 bench.py --steps 3 $Q > $P/bench_prof.log 2>&1 || exit 1
$T 300 rocprofv3 --kernel-trace --stats -d $P/prof_polymul -o vm --output-format csv -- python3 bench.py --workload polymul > $P/polymul_prof.log 2>&1 || exit 1
ONE="python3 bench.py --steps 1 --warmup 0 $Q"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o vm --output-format csv -- $ONE > $P/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o vm --output-format csv -- $ONE > $P/pmc_write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $P/pmc_sq -o vm --output-format csv -- $ONE > $P/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d $P/pmc_act -o vm --output-format csv -- $ONE > $P/pmc_act.log 2>&1 || exit 1
ONE128="python3 bench.py --level 128 --steps 1 --warmup 0 $Q"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $P/pmc_sq128 -o vm --output-format csv -- $ONE128 > $P/pmc_sq128.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d $P/pmc_act128 -o vm --output-format csv -- $ONE128 > $P/pmc_act128.log 2>&1 || exit 1
PM="python3 bench.py --workload polymul --steps 1 --warmup 0"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch_polymul -o vm --output-format csv -- $PM > $P/pmc_fetch_polymul.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $P/pmc_sq_polymul -o vm --output-format csv -- $PM > $P/pmc_sq_polymul.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d $P/pmc_act_polymul -o vm --output-format csv -- $PM > $P/pmc_act_polymul.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write_polymul -o vm --output-format csv -- $PM > $P/pmc_write_polymul.log 2>&1 || exit 1
# cost of one vector / LDS instruction at one and more wavefronts per SIMD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/valu_cost_ubench tools/valu_cost_ubench.hip 2> /dev/null && $T 60 /tmp/valu_cost_ubench > $P/valu_cost_ubench.txt || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/lds_cost_ubench tools/lds_cost_ubench.hip 2> /dev/null && $T 60 /tmp/lds_cost_ubench > $P/lds_cost_ubench.txt || exit 1
# the CSVs of the counter passes are large; keep per-kernel sums only
python3 tools/collect_profiles.py $P --sums-only
rm -rf $P/pmc_*/vm
echo measure done
