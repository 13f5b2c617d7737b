"""Per-phase clock counts of one wavefront of the throughput blind-rotate kernel.

Needs a library built with -DTFHE_STAMPS (diagnostic build, never the shipped one):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-sched-strategy=max-ilp \
        -DTFHE_STAMPS -o /tmp/libtfhe_hip_stamps.so tfhe_amd/csrc/tfhe_hip.hip
  TFHE_HIP_LIB=/tmp/libtfhe_hip_stamps.so python tools/phase_stamps.py
The stamps serialise the phases, so the shares are indicative only.
"""
import ctypes as C, os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tfhe_amd
from bench import synthetic_keys, LEVELS
lv = LEVELS[80]
params = tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"])
eng = tfhe_amd.Engine(0)
K = synthetic_keys(80, 5)
eng.load_keys(params, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
B = 4096
dev = torch.device("cuda", 0)
rnd = lambda shape: torch.randint(-2**31, 2**31, shape, dtype=torch.int64, device=dev).to(torch.int32)
ca_a, ca_b, cb_a, cb_b = rnd((B, 500)), rnd((B,)), rnd((B, 500)), rnd((B,))
out_a = torch.empty((B, 500), dtype=torch.int32, device=dev); out_b = torch.empty((B,), dtype=torch.int32, device=dev)
for _ in range(2):
    eng.gate_dev("nand", ca_a, ca_b, cb_a, cb_b, out_a, out_b)
torch.cuda.synchronize()
st = (C.c_ulonglong * 16)()
eng.lib.tfhe_hip_debug_stamps.argtypes = [C.c_void_p]
print("rc", eng.lib.tfhe_hip_debug_stamps(C.byref(st)))
names = {8: "loop/abar", 0: "rotate", 1: "digits", 2: "fwd", 3: "mirror_store", 4: "pointwise", 5: "inv0", 6: "unpark", 7: "inv1"}
tot = sum(st)
for k, n in names.items():
    print(f"{n:14s} {st[k]:12d}  {100.0*st[k]/tot:5.1f}%  per-iter-level {st[k]/1000.0 if k in (8,0) else st[k]/1000.0:9.1f}")
print("total", tot)
