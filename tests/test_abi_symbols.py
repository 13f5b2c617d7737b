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


def test_key_file_layout_without_a_gpu():
    """tfhe_hip_key_file_size / _pack need no context: header fields, section order and the
    8-byte alignment of the variances."""
    import numpy as np
    from tfhe_amd import _lib
    lib = _lib.load()
    p = _lib.Params(3, 1024, 1, 2, 10, 8, 2, 3)
    rows = 3 * 8 * 4
    bk = np.arange(3 * 2 * 2 * 2 * 1024, dtype=np.int32)
    ks_a = np.arange(rows * 3, dtype=np.int32) + 7
    ks_b = np.arange(rows, dtype=np.int32) - 5
    ks_var = np.linspace(0.0, 1.0, rows)
    size = C.c_size_t()
    assert lib.tfhe_hip_key_file_size(C.byref(p), 1, C.byref(size)) == 0
    want = 64 + bk.size * 4 + ks_a.size * 4 + rows * 4
    want = (want + 7) // 8 * 8 + rows * 8
    assert size.value == want
    out = np.zeros(size.value, np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.tfhe_hip_key_file_pack(C.byref(p), vp(bk), vp(ks_a), vp(ks_b), vp(ks_var), vp(out), out.size) == 0
    assert bytes(out[:8]) == b"TFHEHIPK"
    assert list(np.frombuffer(out[8:16].tobytes(), np.uint32)) == [1, 64]
    assert list(np.frombuffer(out[16:48].tobytes(), np.int32)) == [3, 1024, 1, 2, 10, 8, 2, 3]
    assert int(np.frombuffer(out[56:64].tobytes(), np.uint64)[0]) == want - 64
    o = 64
    assert (np.frombuffer(out[o:o + bk.size * 4].tobytes(), np.int32) == bk).all()
    o += bk.size * 4
    assert (np.frombuffer(out[o:o + ks_a.size * 4].tobytes(), np.int32) == ks_a).all()
    o += ks_a.size * 4
    assert (np.frombuffer(out[o:o + rows * 4].tobytes(), np.int32) == ks_b).all()
    assert (np.frombuffer(out[want - rows * 8:].tobytes(), np.float64) == ks_var).all()
    # too small a buffer, unsupported parameters
    assert lib.tfhe_hip_key_file_pack(C.byref(p), vp(bk), vp(ks_a), vp(ks_b), vp(ks_var), vp(out), 100) != 0
    bad = _lib.Params(3, 512, 1, 2, 10, 8, 2, 3)
    assert lib.tfhe_hip_key_file_size(C.byref(bad), 1, C.byref(size)) != 0
