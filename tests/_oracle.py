"""ctypes binding of oracle/liboracle.so (the CPU checker).

Only tests, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

MUL_EXACT, MUL_FFT, MUL_FFT_PLAN_PER_CALL = 0, 1, 2
GATES = ["nand", "or", "and", "xor", "xnor", "nor", "andny", "andyn", "orny", "oryn", "mux", "not"]
GATE_ID = {g: i for i, g in enumerate(GATES)}


class Params(C.Structure):
    _fields_ = [(f, C.c_int32) for f in
                ("n", "N", "k", "l", "bg_bit", "ks_t", "ks_base_bit", "n_ks")] + \
               [("ks_stdev", C.c_double), ("bk_stdev", C.c_double)]


def _build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle.cpp", "textbook.cpp", "oracle.h")]
    if (not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


_build()
lib = C.CDLL(LIB_PATH)

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
PP = C.POINTER(Params)
DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int32)

lib.orc_params_for.argtypes = [C.c_int, PP]
lib.orc_encode_message.argtypes = [C.c_int32, C.c_int32]
lib.orc_encode_message.restype = C.c_int32
lib.orc_decode_message.argtypes = [C.c_int32, C.c_int32]
lib.orc_decode_message.restype = C.c_int32
lib.orc_f64_to_torus32.argtypes = [C.c_double]
lib.orc_f64_to_torus32.restype = C.c_int32
lib.orc_mul_by_monomial.argtypes = [i32p, C.c_int32, C.c_int32, i32p]
lib.orc_poly_add.argtypes = [i32p, C.c_int32, i32p, C.c_int32, i32p]
lib.orc_poly_sub.argtypes = [i32p, C.c_int32, i32p, C.c_int32, i32p]
lib.orc_poly_multiplier.argtypes = [i32p, C.c_int32, i32p, C.c_int32, i32p, C.c_int, DP]
lib.orc_torus_polynomial_mul_r.argtypes = [i32p, i32p, C.c_int32, i32p, C.c_int32, C.c_int, DP]
lib.orc_decompose.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32, i32p]
lib.orc_extern_mul.argtypes = [PP, i32p, i32p, C.c_void_p, i32p, DP, C.c_int, DP]
lib.orc_blind_rotate.argtypes = [PP, i32p, DP, i32p, i32p, C.c_void_p, C.c_int, DP]
lib.orc_extract_lwe.argtypes = [PP, i32p, i32p, IP]
lib.orc_bootstrap_noks.argtypes = [PP, i32p, C.c_void_p, C.c_int32, i32p, C.c_int32, i32p, IP, DP,
                                   C.c_int, DP]
lib.orc_key_switch.argtypes = [PP, i32p, i32p, C.c_void_p, i32p, C.c_int32, i32p, IP, DP]
lib.orc_bootstrap.argtypes = [PP, i32p, C.c_void_p, i32p, i32p, C.c_void_p, C.c_int32, i32p,
                              C.c_int32, i32p, IP, DP, C.c_int, DP]
lib.orc_gate_prologue.argtypes = [C.c_int, C.c_int32, i32p, C.c_int32, C.c_double, i32p, C.c_int32,
                                  C.c_double, i32p, IP, DP, IP]
lib.orc_gate.argtypes = [PP, C.c_int, i32p, C.c_void_p, i32p, i32p, C.c_void_p,
                         i32p, C.c_int32, C.c_double, i32p, C.c_int32, C.c_double,
                         i32p, C.c_int32, C.c_double, i32p, IP, DP, C.c_int, DP]
lib.orc_generate_keys.argtypes = [PP, C.c_uint64, i32p, i32p, i32p, f64p, i32p, i32p, f64p]
lib.orc_generate_synthetic_keys.argtypes = [PP, C.c_uint64, i32p, f64p, i32p, i32p, f64p]
lib.orc_encrypt.argtypes = [PP, i32p, C.c_int, C.c_uint64, i32p, IP, DP]
lib.orc_lwe_phase.argtypes = [C.c_int32, i32p, C.c_int32, i32p]
lib.orc_lwe_phase.restype = C.c_int32
lib.orc_decrypt.argtypes = [C.c_int32, i32p, C.c_int32, i32p]
lib.orc_fill_uniform_i32.argtypes = [C.c_uint64, i32p, C.c_size_t]
lib.orc_fill_range_i32.argtypes = [C.c_uint64, C.c_int32, C.c_int32, i32p, C.c_size_t]


lib.orc_tb_decompose.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32, i32p]
lib.orc_tb_extern_product.argtypes = [PP, i32p, i32p, i32p]
lib.orc_tb_blind_rotate.argtypes = [PP, i32p, i32p, i32p]
lib.orc_tb_bootstrap_noks.argtypes = [PP, i32p, C.c_int32, i32p, C.c_int32, i32p, IP]
lib.orc_tb_bootstrap.argtypes = [PP, i32p, i32p, i32p, C.c_void_p, C.c_int32, i32p, C.c_int32, i32p, IP, DP]
lib.orc_tb_gate.argtypes = [PP, C.c_int, i32p, i32p, i32p, C.c_void_p, i32p, C.c_int32, i32p, C.c_int32,
                            i32p, C.c_int32, i32p, IP, DP]
lib.orc_tb_generate_keys.argtypes = [PP, C.c_uint64, i32p, i32p, i32p, i32p, i32p, f64p]


def _vp(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


def params(level):
    p = Params()
    assert lib.orc_params_for(level, C.byref(p)) == 0
    return p


def small_params(n=8, l=2, bg_bit=10):
    """Reference parameter shape with a short LWE dimension (N stays 1024)."""
    p = params(80)
    p.n = n
    p.n_ks = n
    p.l = l
    p.bg_bit = bg_bit
    return p


def encode_message(mu, space):
    return lib.orc_encode_message(mu, space)


def decode_message(phase, space):
    return lib.orc_decode_message(int(np.int32(phase)), space)


def mul_by_monomial(p, power):
    p = np.ascontiguousarray(p, dtype=np.int32)
    out = np.empty_like(p)
    lib.orc_mul_by_monomial(p, len(p), power, out)
    return out


def poly_sub(a, b):
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.empty(max(len(a), len(b)), dtype=np.int32)
    lib.orc_poly_sub(a, len(a), b, len(b), out)
    return out


def poly_multiplier(a, b, mode=MUL_EXACT):
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.empty(len(a) + len(b) - 1, dtype=np.int32)
    mf = C.c_double(0.0)
    lib.orc_poly_multiplier(a, len(a), b, len(b), out, mode, C.byref(mf))
    return out, mf.value


def mul_r(result, p1, p2, mode=MUL_EXACT):
    result = np.array(result, dtype=np.int32)
    p1 = np.ascontiguousarray(p1, dtype=np.int32)
    p2 = np.ascontiguousarray(p2, dtype=np.int32)
    mf = C.c_double(0.0)
    lib.orc_torus_polynomial_mul_r(result, p1, len(p1), p2, len(p2), mode, C.byref(mf))
    return result, mf.value


def decompose(poly, l, bg_bit):
    poly = np.ascontiguousarray(poly, dtype=np.int32)
    out = np.empty((l, len(poly)), dtype=np.int32)
    lib.orc_decompose(poly, len(poly), l, bg_bit, out)
    return out


def extern_mul(P, accum, bk_i, bk_i_var=None, mode=MUL_EXACT):
    accum = np.ascontiguousarray(accum, dtype=np.int32)
    bk_i = np.ascontiguousarray(bk_i, dtype=np.int32)
    out = np.empty_like(accum)
    var = C.c_double(0.0)
    mf = C.c_double(0.0)
    lib.orc_extern_mul(C.byref(P), accum, bk_i, _vp(bk_i_var), out, C.byref(var), mode, C.byref(mf))
    return out, var.value, mf.value


def blind_rotate(P, acc, bara, bk, bk_var=None, mode=MUL_EXACT):
    acc = np.array(acc, dtype=np.int32)
    bara = np.ascontiguousarray(bara, dtype=np.int32)
    var = C.c_double(0.0)
    mf = C.c_double(0.0)
    lib.orc_blind_rotate(C.byref(P), acc, C.byref(var), bara, bk, _vp(bk_var), mode, C.byref(mf))
    return acc, var.value, mf.value


def extract_lwe(P, acc):
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    out = np.empty(P.N * P.k, dtype=np.int32)
    b = C.c_int32(0)
    lib.orc_extract_lwe(C.byref(P), acc, out, C.byref(b))
    return out, b.value


def bootstrap_noks(P, bk, mu, a, b, bk_var=None, mode=MUL_EXACT):
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty(P.N * P.k, dtype=np.int32)
    ob = C.c_int32(0)
    var = C.c_double(0.0)
    mf = C.c_double(0.0)
    lib.orc_bootstrap_noks(C.byref(P), bk, _vp(bk_var), mu, a, int(b), out, C.byref(ob),
                           C.byref(var), mode, C.byref(mf))
    return out, ob.value, mf.value


def key_switch(P, ks_a, ks_b, in_a, in_b, ks_var=None):
    in_a = np.ascontiguousarray(in_a, dtype=np.int32)
    out = np.empty(P.n, dtype=np.int32)
    ob = C.c_int32(0)
    var = C.c_double(0.0)
    lib.orc_key_switch(C.byref(P), ks_a, ks_b, _vp(ks_var), in_a, int(in_b), out, C.byref(ob),
                       C.byref(var))
    return out, ob.value, var.value


def bootstrap(P, bk, ks_a, ks_b, mu, a, b, bk_var=None, ks_var=None, mode=MUL_EXACT):
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty(P.n, dtype=np.int32)
    ob = C.c_int32(0)
    var = C.c_double(0.0)
    mf = C.c_double(0.0)
    lib.orc_bootstrap(C.byref(P), bk, _vp(bk_var), ks_a, ks_b, _vp(ks_var), mu, a, int(b), out,
                      C.byref(ob), C.byref(var), mode, C.byref(mf))
    return out, ob.value, var.value, mf.value


def gate_prologue(gate, ca, cb):
    """ca, cb: (a, b, var).  Returns (x_a, x_b, x_var, mu)."""
    n = len(ca[0])
    xa = np.empty(n, dtype=np.int32)
    xb = C.c_int32(0)
    xv = C.c_double(0.0)
    mu = C.c_int32(0)
    rc = lib.orc_gate_prologue(GATE_ID[gate], n, np.ascontiguousarray(ca[0], dtype=np.int32),
                               int(ca[1]), float(ca[2]), np.ascontiguousarray(cb[0], dtype=np.int32),
                               int(cb[1]), float(cb[2]), xa, C.byref(xb), C.byref(xv), C.byref(mu))
    assert rc == 0
    return xa, xb.value, xv.value, mu.value


def gate(P, gate, keys, ca, cb, cc=None, mode=MUL_EXACT):
    """keys: dict with bk, bk_var, ks_a, ks_b, ks_var.  Samples are (a, b, var)."""
    if cc is None:
        cc = (np.zeros(P.n, dtype=np.int32), 0, 0.0)
    out = np.empty(P.n, dtype=np.int32)
    ob = C.c_int32(0)
    var = C.c_double(0.0)
    mf = C.c_double(0.0)
    rc = lib.orc_gate(C.byref(P), GATE_ID[gate], keys["bk"], _vp(keys.get("bk_var")), keys["ks_a"],
                      keys["ks_b"], _vp(keys.get("ks_var")),
                      np.ascontiguousarray(ca[0], dtype=np.int32), int(ca[1]), float(ca[2]),
                      np.ascontiguousarray(cb[0], dtype=np.int32), int(cb[1]), float(cb[2]),
                      np.ascontiguousarray(cc[0], dtype=np.int32), int(cc[1]), float(cc[2]),
                      out, C.byref(ob), C.byref(var), mode, C.byref(mf))
    assert rc == 0
    return out, ob.value, var.value


def key_shapes(P):
    base = 1 << P.ks_base_bit
    return dict(
        lwe_key=(P.n,), tlwe_key=(P.k + 1, P.N),
        bk=(P.n, P.l, P.k + 1, P.k + 1, P.N), bk_var=(P.n, P.l, P.k + 1),
        ks_a=(P.n_ks, P.ks_t, base, P.n), ks_b=(P.n_ks, P.ks_t, base),
        ks_var=(P.n_ks, P.ks_t, base))


def generate_keys(P, seed):
    sh = key_shapes(P)
    k = {name: np.zeros(shape, dtype=np.float64 if name.endswith("var") else np.int32)
         for name, shape in sh.items()}
    lib.orc_generate_keys(C.byref(P), seed, k["lwe_key"], k["tlwe_key"], k["bk"], k["bk_var"],
                          k["ks_a"], k["ks_b"], k["ks_var"])
    return k


def generate_synthetic_keys(P, seed):
    sh = key_shapes(P)
    k = {name: np.zeros(sh[name], dtype=np.float64 if name.endswith("var") else np.int32)
         for name in ("bk", "bk_var", "ks_a", "ks_b", "ks_var")}
    lib.orc_generate_synthetic_keys(C.byref(P), seed, k["bk"], k["bk_var"], k["ks_a"], k["ks_b"],
                                    k["ks_var"])
    return k


def encrypt(P, lwe_key, message, seed):
    a = np.empty(P.n, dtype=np.int32)
    b = C.c_int32(0)
    var = C.c_double(0.0)
    lib.orc_encrypt(C.byref(P), lwe_key, int(bool(message)), seed, a, C.byref(b), C.byref(var))
    return a, b.value, var.value


def decrypt(a, b, lwe_key):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return bool(lib.orc_decrypt(len(a), a, int(b), lwe_key))


def uniform_i32(seed, shape):
    out = np.empty(shape, dtype=np.int32)
    lib.orc_fill_uniform_i32(seed, out.reshape(-1), out.size)
    return out


def range_i32(seed, lo, hi, shape):
    out = np.empty(shape, dtype=np.int32)
    lib.orc_fill_range_i32(seed, lo, hi, out.reshape(-1), out.size)
    return out


# ---- the textbook-correct variant (oracle/textbook.cpp) ------------------------------------------

def tb_params(level=80, n=None, bk_stdev=None):
    """Parameter shape of Parameters::with(level) for the textbook variant: the key-switch key
    has N k rows.  n: a shorter LWE dimension for fast tests."""
    p = params(level)
    if n is not None:
        p.n = n
    p.n_ks = p.N * p.k
    if bk_stdev is not None:
        p.bk_stdev = bk_stdev
    return p


def tb_key_shapes(P):
    base = 1 << P.ks_base_bit
    return dict(lwe_key=(P.n,), tlwe_key=(P.k, P.N), bk=(P.n, P.l, P.k + 1, P.k + 1, P.N),
                ks_a=(P.n_ks, P.ks_t, base, P.n), ks_b=(P.n_ks, P.ks_t, base), ks_var=(P.n_ks, P.ks_t, base))


def tb_generate_keys(P, seed):
    k = {name: np.zeros(shape, dtype=np.float64 if name.endswith("var") else np.int32)
         for name, shape in tb_key_shapes(P).items()}
    lib.orc_tb_generate_keys(C.byref(P), seed, k["lwe_key"], k["tlwe_key"].reshape(-1), k["bk"].reshape(-1),
                             k["ks_a"].reshape(-1), k["ks_b"].reshape(-1), k["ks_var"].reshape(-1))
    return k


def tb_decompose(poly, l, bg_bit):
    poly = np.ascontiguousarray(poly, dtype=np.int32)
    out = np.empty((l, len(poly)), dtype=np.int32)
    lib.orc_tb_decompose(poly, len(poly), l, bg_bit, out.reshape(-1))
    return out


def tb_extern_product(P, acc, bk_i):
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    out = np.empty_like(acc)
    lib.orc_tb_extern_product(C.byref(P), acc.reshape(-1), np.ascontiguousarray(bk_i).reshape(-1), out.reshape(-1))
    return out


def tb_blind_rotate(P, acc, bara, bk):
    acc = np.array(acc, dtype=np.int32)
    lib.orc_tb_blind_rotate(C.byref(P), acc.reshape(-1), np.ascontiguousarray(bara, dtype=np.int32),
                            np.ascontiguousarray(bk).reshape(-1))
    return acc


def tb_bootstrap_noks(P, bk, mu, a, b):
    out = np.empty(P.N * P.k, dtype=np.int32)
    ob = C.c_int32(0)
    lib.orc_tb_bootstrap_noks(C.byref(P), np.ascontiguousarray(bk).reshape(-1), mu,
                              np.ascontiguousarray(a, dtype=np.int32), int(b), out, C.byref(ob))
    return out, ob.value


def tb_gate(P, gate, keys, ca, cb, cc=None):
    """Samples are (a, b); returns (a, b, variance of the key switch)."""
    if cc is None:
        cc = (np.zeros(P.n, dtype=np.int32), 0)
    out = np.empty(P.n, dtype=np.int32)
    ob = C.c_int32(0)
    var = C.c_double(0.0)
    rc = lib.orc_tb_gate(C.byref(P), GATE_ID[gate], np.ascontiguousarray(keys["bk"]).reshape(-1),
                         np.ascontiguousarray(keys["ks_a"]).reshape(-1),
                         np.ascontiguousarray(keys["ks_b"]).reshape(-1), _vp(keys.get("ks_var")),
                         np.ascontiguousarray(ca[0], dtype=np.int32), int(ca[1]),
                         np.ascontiguousarray(cb[0], dtype=np.int32), int(cb[1]),
                         np.ascontiguousarray(cc[0], dtype=np.int32), int(cc[1]), out, C.byref(ob), C.byref(var))
    assert rc == 0
    return out, ob.value, var.value
