"""The C-ABI library builds, loads and exports every symbol the header declares.
No compute call is made (runs without a GPU)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_hip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tfhe_amd import _lib, build
    build.build_library()
    names = _declared()
    assert len(names) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == names


def test_create_fails_loudly_without_gpu_or_bad_device():
    from tfhe_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.tfhe_hip_create(C.byref(ctx), 4096)   # no such device anywhere
    assert rc != 0 and not ctx.value


def test_key_blob_sizes():
    from tfhe_amd import _lib
    lib = _lib.load()
    p = _lib.Params(500, 1024, 1, 2, 10, 8, 2, 500)
    a, b = C.c_size_t(), C.c_size_t()
    assert lib.tfhe_hip_key_blob_sizes(C.byref(p), C.byref(a), C.byref(b)) == 0
    assert a.value == 500 * 2 * 2 * 2048 * 16          # 64 MB of FFT-domain key
    assert b.value == 500 * 8 * 4 * (504 * 4 + 8)
    bad = _lib.Params(500, 512, 1, 2, 10, 8, 2, 500)
    assert lib.tfhe_hip_key_blob_sizes(C.byref(bad), C.byref(a), C.byref(b)) != 0
