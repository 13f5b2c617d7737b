"""Opcode histogram of one kernel in `hipcc -S` output.
usage: python tools/isa_histogram.py <file.s> <substring of the kernel's mangled name> [...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()


def kernel_body(name):
    m = re.search(r"^(_Z[^\n:]*" + re.escape(name) + r"[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
    return m.group(2) if m else None


for name in sys.argv[2:]:
    b = kernel_body(name)
    if b is None:
        print(name, "not found")
        continue
    ops = collections.Counter()
    for line in b.splitlines():
        line = line.strip()
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    groups = collections.Counter()
    for k, v in ops.items():
        if k.startswith("ds_"):
            groups["lds:" + k] += v
        elif k.startswith("v_permlane"):
            groups["permlane"] += v
        elif k.startswith("v_accvgpr"):
            groups["accvgpr"] += v
        elif k.startswith("s_waitcnt"):
            groups["waitcnt"] += v
        elif k.startswith("s_nop"):
            groups["s_nop"] += v
        elif k.startswith("global_") or k.startswith("scratch_") or k.startswith("buffer_"):
            groups[k] += v
        elif k.startswith("s_"):
            groups["scalar"] += v
        elif k.endswith("_f64") and not k.startswith("v_cvt"):
            groups["f64"] += v
        else:
            groups["v_other:" + k] += v
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    print(name, "total", sum(ops.values()), "valu", valu)
    for k, v in sorted(groups.items(), key=lambda x: -x[1])[:28]:
        print("    %-34s %d" % (k, v))
