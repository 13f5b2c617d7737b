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
    # max-ilp scheduling: one wavefront per SIMD has nobody to hide a dependent
    # instruction's latency behind, so independent work is interleaved by the compiler
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


HOST_BIN = os.path.join(HERE, "host", "bin")


def build_host_programs():
    """Builds the C++ host-API example and its parity test (g++; they link the library)."""
    root = os.path.dirname(HERE)
    os.makedirs(HOST_BIN, exist_ok=True)
    common = ["g++", "-O2", "-std=c++17", "-fwrapv", "-Wall", "-Wextra"]
    link = ["-L" + HERE, "-ltfhe_hip", "-Wl,-rpath,$ORIGIN/../.."]
    jobs = [
        ([os.path.join(root, "examples", "homomorphic_xor.cpp")], "homomorphic_xor"),
        ([os.path.join(root, "tests", "host_api_test.cpp"), os.path.join(root, "oracle", "oracle.cpp")],
         "host_api_test"),
    ]
    hdr = os.path.join(HERE, "host", "tfhe.hpp")
    for srcs, name in jobs:
        out = os.path.join(HOST_BIN, name)
        deps = srcs + [hdr, OUT]
        if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
            continue
        subprocess.check_call(common + srcs + ["-o", out] + link)
    return HOST_BIN


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
