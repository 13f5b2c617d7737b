"""Builds variants of the kernel unit with different tuning macros (pipeline depths, group
sizes, scheduler strategy, asm/C++ butterflies) as tools/_sweep/lib<name>.so, in parallel, and
lists those that built in tools/_sweep/names.txt (the scratch size of the Bit80 kernel is printed:
a few bytes saved around the ciphertext loop are harmless, spills inside it are not).  On the GPU box each variant is
then timed with
    TFHE_HIP_LIB=tools/_sweep/lib<name>.so python bench.py --steps 3 --no-cpu-baseline --no-latency --no-bit128
by tools/sweep_run.sh (results: profiles/r02_sweep_*.txt).
"""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_sweep")
base = dict(TWG=8, TWISTG=8, K0D=3, K1D=2, PB=1, TC=0, DIF=0, DIT=1, FD=1, ACC=1, S="max-ilp")
singles = [dict(TWG=4), dict(TWG=16), dict(TWISTG=4), dict(TWISTG=16), dict(K0D=2), dict(K0D=4), dict(K1D=1),
           dict(K1D=3), dict(K1D=4), dict(PB=0), dict(TC=4), dict(TC=8), dict(TC=16), dict(DIF=1), dict(DIT=0),
           dict(FD=0), dict(ACC=0), dict(S="default"), dict(S="max-memory-clause"), dict(FD=0, ACC=0),
           dict(TC=8, PB=0), dict(K0D=4, K1D=4), dict(TWG=16, TWISTG=16), dict(K0D=2, K1D=2), dict(TWISTG=16, TC=8)]
variants = [dict(base)] + [dict(base, **e) for e in singles]


def name(d):
    diff = [f"{k}{str(v)[0] if k == 'S' else v}" for k, v in d.items() if base[k] != v]
    return "s_" + "_".join(diff) if diff else "s_base"


def build(d):
    n = name(d)
    flags = [f"-DWFFT_TW_GROUP={d['TWG']}", f"-DWFFT_TWIST_GROUP={d['TWISTG']}", f"-DTFHE_K0_DEPTH={d['K0D']}",
             f"-DTFHE_K1_DEPTH={d['K1D']}", f"-DWFFT_PIPE_BARRIER={d['PB']}",
             f"-DWFFT_TAIL_CHUNK={d['TC']}", f"-DWFFT_ASM_DIF={d['DIF']}", f"-DWFFT_ASM_DIT={d['DIT']}",
             f"-DWFFT_FWD_DIT={d['FD']}", f"-DTFHE_ACC_VGPR={d['ACC']}"]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-c", "--offload-arch=gfx950"] + flags
    if d["S"] != "default":
        cmd += ["-mllvm", f"-amdgpu-sched-strategy={d['S']}"]
    obj = os.path.join(OUT, n + ".o")
    r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage", "-o", obj,
                              os.path.join(ROOT, "tfhe_amd/csrc/tfhe_hip.hip")], capture_output=True, text=True)
    if r.returncode:
        return n, "FAIL", r.stderr[-300:]
    m = re.search(r"blind_rotate_kernelILi2ELi10ELb0.*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, re.S)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                    os.path.join(OUT, f"lib{n}.so"), obj, os.path.join(ROOT, "tfhe_amd/csrc/host_api.o"),
                    "-ldl", "-lpthread"], check=True)
    os.remove(obj)
    return n, "ok", m.group(1) if m else "?"


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(7) as ex:
        res = list(ex.map(build, variants))
    for r in res:
        print(*r)
    open(os.path.join(OUT, "names.txt"), "w").write("\n".join(r[0] for r in res if r[1] == "ok"))
