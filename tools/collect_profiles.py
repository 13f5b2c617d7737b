"""Copies the summaries of one measurement pass into profiles/ (kernel stats, bench lines, PMC sums
per kernel, traffic).  The pass directory holds bench80.json, bench128.json, bench80_refkeys.json,
add32.json, polymul.json (bench.py lines), prof/ (rocprofv3 --kernel-trace --stats --output-format csv),
pmc_fetch/, pmc_write/, pmc_sq/ (rocprofv3 --pmc ... --kernel-trace --output-format csv of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency`) and bench_prof.log.
usage: python tools/collect_profiles.py gpurun_out/<pass>"""
import collections, csv, glob, json, shutil, sys
R, P = sys.argv[1], "profiles"
names = {"bench80": "r01_bench_bit80", "bench128": "r01_bench_bit128", "bench80_refkeys": "r01_bench_bit80_reference_style_keys",
         "add32": "r01_bench_add32", "polymul": "r01_bench_polymul"}
for f, o in names.items():
    d = json.load(open(f"{R}/{f}.json"))
    print(f, round(d["value"], 1), d.get("p50_single_gate_latency_ms"), d.get("roofline", {}).get("frac"),
          d.get("kernels_ms_per_launch"), (d.get("cpu_baseline") or {}).get("value"))
    shutil.copy(f"{R}/{f}.json", f"{P}/{o}.json")
ks = glob.glob(R + "/prof/vm/*_kernel_stats.csv")[0]
for i, row in enumerate(csv.DictReader(open(ks))):
    if i < 3: print(row["Name"][:60], row["Calls"], row["AverageNs"], row["Percentage"])
shutil.copy(ks, P + "/r01_bench_kernel_stats.csv")
shutil.copy(R + "/bench_prof.log", P + "/r01_bench_under_rocprof.log")
summ = {}
for kind in ("fetch", "write", "sq"):
    f = glob.glob(f"{R}/pmc_{kind}/vm/*_counter_collection.csv")[0]
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        k = ("blind_rotate_kernel<2,10,false>" if "blind_rotate" in kn else "keyswitch_tile_kernel<8>" if "keyswitch" in kn
             else "prologue_kernel" if "prologue" in kn else None)
        if k is None: continue
        agg[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    out = {}
    for (k, c), v in agg.items(): out.setdefault(k, {})[c] = {"sum": v, "dispatches": cnt[(k, c)]}
    summ[kind] = out
    json.dump({"command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 1 "
                          "--warmup 0 --no-cpu-baseline --no-latency", "per_kernel": out},
              open(f"{P}/r01_pmc_{kind}_summary.json", "w"), indent=1)
br = "blind_rotate_kernel<2,10,false>"
f, w = summ["fetch"][br]["FETCH_SIZE"]["sum"], summ["write"][br]["WRITE_SIZE"]["sum"]
traffic = 2 * f * 1024 + w * 1024
json.dump({"kernel": br, "workload": "bench.py default: 16384 NAND gates, Bit80", "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
           "correction": "FETCH_SIZE x2 on gfx950, units KiB", "traffic_bytes_per_launch": traffic},
          open(P + "/r01_traffic.json", "w"), indent=1)
print("traffic GB", traffic / 1e9, {k: v["sum"] for k, v in summ["sq"][br].items()})
