"""rust/src/ffi.rs (source only, never compiled here) declares exactly the symbols of
include/tfhe_hip.h, with as many parameters each."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    text = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(tfhe_hip_[a-z_]+)\s*\(([^;]*)\);", text):
        out[name] = len([a for a in args.split(",") if a.strip() and a.strip() != "void"])
    return out


def _rust():
    text = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    out = {}
    for name, args in re.findall(r"pub fn (tfhe_hip_[a-z_]+)\(([^;]*?)\)\s*(?:->[^;]*)?;", text, flags=re.S):
        out[name] = len([a for a in args.split(",") if a.strip()])
    return out


def test_ffi_rs_declares_every_header_symbol():
    h, r = _header(), _rust()
    assert sorted(h) == sorted(r)
    for name in h:
        assert h[name] == r[name], name


def test_gpu_rs_calls_only_declared_symbols():
    text = open(os.path.join(ROOT, "rust", "src", "gpu.rs")).read()
    code = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("//"))
    used = set(re.findall(r"\b(tfhe_hip_[a-z_]+)\(", code))
    assert used and used <= set(_rust())
