"""Host buffers in, host buffers out: tfhe_hip_gate on 16 384 NAND gates through the context's pinned
staging pool, against the same batch with operands resident in HBM (tfhe_hip_gate_dev).  The
PCIe-inclusive rate is never the benchmark's `value`; it is what a caller of the host-pointer entry
points sees.   usage: python tools/pcie_inclusive.py [80|128]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tfhe_amd  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 80
lv = bench.LEVELS[level]
B = 16384
eng = tfhe_amd.Engine(0)
eng.use_torch_stream()
K = bench.synthetic_keys(level, 3)
eng.load_keys(tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"]), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
rng = np.random.default_rng(5)
u = lambda shape: rng.integers(-2**31, 2**31, size=shape, dtype=np.int64).astype(np.int32)
ca, cb = (u((B, lv["n"])), u((B,))), (u((B, lv["n"])), u((B,)))
dev = torch.device("cuda", 0)
d = [torch.from_numpy(x).to(dev) for x in (ca[0], ca[1], cb[0], cb[1])]
out_a = torch.empty((B, lv["n"]), dtype=torch.int32, device=dev)
out_b = torch.empty((B,), dtype=torch.int32, device=dev)


def best(f, reps=6):
    t = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return min(t[1:]) * 1e3


host = best(lambda: eng.gate("nand", ca, cb))
resident = best(lambda: eng.gate_dev("nand", d[0], d[1], d[2], d[3], out_a, out_b))
print(f"Bit{level}, {B} NAND gates: host pointers {host:.1f} ms = {B / host:.1f} K gates/s; "
      f"device resident {resident:.1f} ms = {B / resident:.1f} K gates/s; staging costs {host - resident:.1f} ms "
      f"({100 * (host - resident) / host:.1f} %)")
eng.close()
