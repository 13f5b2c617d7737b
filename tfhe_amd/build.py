"""Builds libtfhe_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tfhe_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "wave_fft.hpp"),
        os.path.join(os.path.dirname(HERE), "include", "tfhe_hip.h")]
OUT = os.path.join(HERE, "libtfhe_hip.so")


def build_library(force=False, verbose=False):
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in DEPS)):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
