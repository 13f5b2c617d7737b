"""Builds variants of the kernel unit, one per "name=-DMACRO=.. -DMACRO2=.." argument, as
tools/_sweep/lib<name>.so (in parallel) and lists those that built in tools/_sweep/names.txt.
The resource usage of the kernels whose mangled names contain --report is printed.
usage: python tools/sweep_variants.py --report polymul64_kernelILb0ELb0 base= a=-DX=1 "b=-DX=2 -DY=3"
On the GPU box: tools/sweep_run.sh (gates) or tools/sweep_run_polymul.sh (isolated products)."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_sweep")
args = sys.argv[1:]
report = []
while args and args[0] == "--report":
    report.append(args[1])
    args = args[2:]
specs = [a.split("=", 1) for a in args]


def build(spec):
    n, flags = spec
    # "@sched=<strategy>" among the flags replaces the default scheduling strategy (max-ilp);
    # "@sched=default" drops the option
    sched = "max-ilp"
    rest = []
    for f in flags.split():
        if f.startswith("@sched="):
            sched = f[7:]
        else:
            rest.append(f)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-c", "--offload-arch=gfx950"]
    if sched != "default":
        cmd += ["-mllvm", "-amdgpu-sched-strategy=" + sched]
    cmd += rest
    obj = os.path.join(OUT, n + ".o")
    r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage", "-o", obj,
                              os.path.join(ROOT, "tfhe_amd/csrc/tfhe_hip.hip")], capture_output=True, text=True)
    if r.returncode:
        return n, "FAIL", r.stderr[-400:]
    info = []
    for k in report:
        m = re.search(re.escape(k) + r".*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                      r"Occupancy \[waves/SIMD\]: (\d+)", r.stderr, re.S)
        info.append(f"{k}: vgpr {m.group(1)} agpr {m.group(2)} scratch {m.group(3)} occ {m.group(4)}" if m else k + ": ?")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                    os.path.join(OUT, f"lib{n}.so"), obj, os.path.join(ROOT, "tfhe_amd/csrc/host_api.o"),
                    "-ldl", "-lpthread"], check=True)
    os.remove(obj)
    return n, "ok", "; ".join(info)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from tfhe_amd import build as _build
    _build.build_library()                      # host_api.o must match ctx.hpp
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    with ThreadPoolExecutor(7) as ex:
        res = list(ex.map(build, specs))
    for r in res:
        print(*r)
    open(os.path.join(OUT, "names.txt"), "w").write("\n".join(r[0] for r in res if r[1] == "ok"))
