"""Summaries of one measurement pass (tools/measure.sh) for profiles/.

  python tools/collect_profiles.py <pass dir> --sums-only   (on the GPU box) reduces every
      counter CSV of the pass to per-kernel sums, <pass dir>/pmc_sums.json
  python tools/collect_profiles.py <pass dir> <prefix>      copies bench lines, kernel statistics,
      counter sums and the derived HBM traffic into profiles/<prefix>_*
Traffic: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide
reads, so it is doubled (the CDNA4 guide's correction).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1]


def kernel_key(name):
    for k in ("blind_rotate_split64_kernel", "blind_rotate64_kernel", "blind_rotate_kernel", "keyswitch_tile_kernel", "keyswitch_kernel",
              "prologue_kernel", "polymul64_kernel", "polymul2_kernel", "polymul_kernel", "bk_fft_kernel"):
        if k in name:
            return k
    return None


if sys.argv[2] == "--sums-only":
    out = {}
    for d in sorted(glob.glob(R + "/pmc_*")):
        files = glob.glob(d + "/vm/*_counter_collection.csv") + glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
        if not os.path.isdir(d) or not files:
            continue
        agg = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for row in csv.DictReader(open(files[0])):
            k = kernel_key(row["Kernel_Name"])
            if k is None:
                continue
            agg[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row.get("Dispatch_Id"))
        per = {}
        for (k, c), v in agg.items():
            per.setdefault(k, {})[c] = {"sum": v, "dispatches": len(disp[(k, c)])}
        out[os.path.basename(d)] = per
    json.dump(out, open(R + "/pmc_sums.json", "w"), indent=1)
    sys.exit(0)

PRE, P = sys.argv[2], "profiles"
for f in ("bench80", "bench128", "bench80_refkeys", "add32", "polymul"):
    d = json.load(open(f"{R}/{f}.json"))
    print(f, round(d["value"], 1), d.get("p50_single_gate_latency_ms"), d.get("roofline", {}).get("frac"),
          d.get("kernels_ms_per_launch"), (d.get("cpu_baseline") or {}).get("value"))
    shutil.copy(f"{R}/{f}.json", f"{P}/{PRE}_{f}.json")
for sub, o in (("prof", "bench_kernel_stats"), ("prof_polymul", "polymul_kernel_stats")):
    ks = glob.glob(f"{R}/{sub}/**/*_kernel_stats.csv", recursive=True)[0]
    for i, row in enumerate(csv.DictReader(open(ks))):
        if i < 3:
            print(row["Name"][:60], row["Calls"], row["AverageNs"], row["Percentage"])
    shutil.copy(ks, f"{P}/{PRE}_{o}.csv")
shutil.copy(R + "/bench_prof.log", f"{P}/{PRE}_bench_under_rocprof.log")
if os.path.exists(R + "/strong_1M_gates.json"):
    shutil.copy(R + "/strong_1M_gates.json", f"{P}/{PRE}_bench_strong_1M_gates.json")
for name in ("valu_cost_ubench.txt", "lds_cost_ubench.txt", "batch_sizes.txt"):
    if os.path.exists(R + "/" + name):
        shutil.copy(R + "/" + name, f"{P}/{PRE}_{name}")
sums = json.load(open(R + "/pmc_sums.json"))
json.dump({"commands": "see tools/measure.sh: one rocprofv3 --pmc run per counter group over "
                       "`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-bit128` "
                       "(and `--workload polymul --steps 1 --warmup 0`)", "per_pass": sums},
          open(f"{P}/{PRE}_pmc_sums.json", "w"), indent=1)


def traffic(fetch_pass, write_pass, kernel):
    f = sums[fetch_pass][kernel]["FETCH_SIZE"]
    w = sums[write_pass][kernel]["WRITE_SIZE"]
    return {"kernel": kernel, "FETCH_SIZE_KiB_raw": f["sum"], "WRITE_SIZE_KiB": w["sum"], "dispatches": f["dispatches"],
            "correction": "FETCH_SIZE x2 on gfx950, units KiB",
            "traffic_bytes_per_launch": (2 * f["sum"] + w["sum"]) * 1024 / f["dispatches"]}


bk = "blind_rotate64_kernel" if "blind_rotate64_kernel" in sums["pmc_fetch"] else "blind_rotate_kernel"
t = traffic("pmc_fetch", "pmc_write", bk)
t["workload"] = "bench.py default: 16384 NAND gates, Bit80"
json.dump(t, open(f"{P}/{PRE}_traffic.json", "w"), indent=1)
pk = next(k for k in ("polymul64_kernel", "polymul2_kernel", "polymul_kernel") if k in sums["pmc_fetch_polymul"])
tp = traffic("pmc_fetch_polymul", "pmc_write_polymul", pk)
tp["workload"] = "bench.py --workload polymul: 65536 products"
json.dump(tp, open(f"{P}/{PRE}_traffic_polymul.json", "w"), indent=1)
print("traffic GB per launch: blind rotate", t["traffic_bytes_per_launch"] / 1e9, "polymul", tp["traffic_bytes_per_launch"] / 1e9)
