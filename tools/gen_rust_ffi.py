"""Regenerates the `extern "C"` block of rust/src/ffi.rs from include/tfhe_hip.h (one declaration
per exported symbol, same order).  usage: python tools/gen_rust_ffi.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
txt = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
protos = re.findall(r"\n((?:int|void|size_t|const char\*)\s+tfhe_hip_[a-z_0-9]+\s*\([^;]*\);)", txt)
SCALAR = {"int": "c_int", "size_t": "usize", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64",
          "uint8_t": "u8", "double": "f64"}
OPAQUE = {"tfhe_hip_ctx": "TfheHipCtx", "tfhe_hip_samples": "TfheHipSamples",
          "tfhe_hip_params": "TfheHipParams", "char": "c_char", "void": "c_void"}


def rtype(c):
    c = c.strip()
    const = c.startswith("const ")
    c2 = c.replace("const ", "").strip()
    ptr = c2.count("*")
    base = c2.replace("*", "").strip()
    b = OPAQUE.get(base, SCALAR.get(base, base))
    for level in range(ptr):
        b = ("*const " if const and level == 0 else "*mut ") + b
    return b


out = []
for p in protos:
    p = " ".join(p.split())
    ret, name, args = re.match(r"(.*?)\s*(tfhe_hip_[a-z_0-9]+)\s*\((.*)\);", p).groups()
    ra = []
    for a in args.split(","):
        a = a.strip()
        if a == "void" or not a:
            continue
        ty, an, arr = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)(\[\d*\])?$", a).groups()
        ty = ty.strip() + ("*" if arr else "")
        if an in ("type", "in", "ref", "box", "fn", "mod", "match"):
            an += "_"
        ra.append(f"{an}: {rtype(ty)}")
    r = ret.strip()
    rets = "" if r == "void" else " -> " + {"int": "c_int", "size_t": "usize", "const char*": "*const c_char"}[r]
    line = f"    pub fn {name}(" + ", ".join(ra) + ")" + rets + ";"
    if len(line) > 100:
        line = f"    pub fn {name}(\n        " + ",\n        ".join(ra) + ",\n    )" + rets + ";"
    out.append(line)
path = os.path.join(ROOT, "rust", "src", "ffi.rs")
src = open(path).read()
head = src[:src.index('extern "C" {')]
open(path, "w").write(head + 'extern "C" {\n' + "\n".join(out) + "\n}\n")
print(len(out), "declarations")
