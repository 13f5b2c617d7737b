"""Per-phase clock counts (s_memtime ticks) of one wavefront of an instrumented kernel.

Needs a library built with -DTFHE_STAMPS (diagnostic build, never the shipped one):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -mllvm -amdgpu-sched-strategy=max-ilp \
        -DTFHE_STAMPS -o /tmp/stamps.o tfhe_amd/csrc/tfhe_hip.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libstamps.so /tmp/stamps.o tfhe_amd/csrc/host_api.o -ldl -lpthread
  TFHE_HIP_LIB=/tmp/libstamps.so python tools/phase_stamps.py
Instrumented: blind_rotate_kernel (mode batch); the kernels with several wavefronts per SIMD are
not (interleaved wavefronts make per-phase shares meaningless; tools/sweep_variants.py with the
ablation macros is the tool for them).
The stamps are not scheduling barriers: instructions move across them, so the shares of
neighbouring phases are indicative only.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import tfhe_amd
from bench import LEVELS, synthetic_keys

which = "batch"
eng = tfhe_amd.Engine(0)
dev = torch.device("cuda", 0)
rnd = lambda shape, lo=-2**31, hi=2**31: torch.randint(lo, hi, shape, dtype=torch.int64, device=dev).to(torch.int32)


def stamps():
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    eng.lib.tfhe_hip_debug_stamps.argtypes = [C.c_void_p]
    assert eng.lib.tfhe_hip_debug_stamps(C.byref(st)) == 0
    return list(st)


def show(names, st, per):
    tot = sum(st[k] for k in names)
    for k, n in names.items():
        print(f"  {n:22s} {st[k] / per:9.0f} ticks  {100.0 * st[k] / max(tot, 1):5.1f}%")
    print(f"  total {tot / per:.0f} ticks per unit")


def main():
    lv = LEVELS[80]
    params = tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"])
    K = synthetic_keys(80, 5)
    eng.load_keys(params, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
    B = 4096
    eng.set_blind_rotate_mode(which)
    ca_a, ca_b, cb_a, cb_b = rnd((B, 500)), rnd((B,)), rnd((B, 500)), rnd((B,))
    out_a = torch.empty((B, 500), dtype=torch.int32, device=dev)
    out_b = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(2):
        eng.gate_dev("nand", ca_a, ca_b, cb_a, cb_b, out_a, out_b)
    st = stamps()
    # the wavefront that reports has finished three ciphertexts of 500 iterations
    print("blind_rotate_kernel, per iteration (two levels)")
    show({8: "loop, abar", 0: "rotate", 1: "digits", 2: "forward", 4: "park, product A",
          5: "inverse A", 6: "unpark, product B", 7: "inverse B"}, st, 1500)


if __name__ == "__main__":
    main()
