"""Blind-rotation time of one bootstrapped NAND batch by batch size and GPU mapping (Bit80 by
default), to place the thresholds of the automatic mode.
usage: python tools/time_batch_sizes.py [--level 80|128] [sizes ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tfhe_amd  # noqa: E402

args = sys.argv[1:]
level = 80
if args and args[0] == "--level":
    level = int(args[1])
    args = args[2:]
sizes = [int(x) for x in args] or [256, 512, 600, 768, 1024, 1100, 1536, 2048, 2100, 2560, 3072, 3500, 4096, 5000]
lv = bench.LEVELS[level]
dev = torch.device("cuda", 0)
eng = tfhe_amd.Engine(0)
eng.use_torch_stream()
K = bench.synthetic_keys(level, 7)
eng.load_keys(tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"]), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
g = torch.Generator(device=dev)
g.manual_seed(1)
B = max(sizes)
rnd = lambda shape: torch.randint(-2**31, 2**31, shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32)
ca_a, ca_b, cb_a, cb_b = rnd((B, lv["n"])), rnd((B,)), rnd((B, lv["n"])), rnd((B,))
out_a = torch.empty((B, lv["n"]), dtype=torch.int32, device=dev)
out_b = torch.empty((B,), dtype=torch.int32, device=dev)
print(f"Bit{level}: milliseconds per batch (whole gate: prologue + blind rotation + key switch), best of 3")
print("%8s %10s %10s %10s %10s" % ("gates", "auto", "batch64", "batch", "split"))
for b in sizes:
    row = []
    for mode in ("auto", "batch64", "batch", "split"):
        if mode == "split" and b > 1024:
            row.append(float("nan"))
            continue
        eng.set_blind_rotate_mode(mode)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.gate_dev("nand", ca_a[:b], ca_b[:b], cb_a[:b], cb_b[:b], out_a[:b], out_b[:b])
            torch.cuda.synchronize()
            if rep:
                best = min(best, time.perf_counter() - t0)
        row.append(best * 1e3)
    print("%8d %10.2f %10.2f %10.2f %10.2f" % (b, *row), flush=True)
eng.close()
