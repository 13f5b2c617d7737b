"""ctypes loader for libtfhe_hip.so (the C ABI declared in include/tfhe_hip.h).

There is no fallback: if the shared library is missing or fails to load, the
import raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``tfhe_amd.build.build_library()``).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# TFHE_HIP_LIB overrides the library path (kernel experiments).
LIB_PATH = os.environ.get("TFHE_HIP_LIB") or os.path.join(HERE, "libtfhe_hip.so")
# the textbook-correct variant, a library of its own with the same entry points
TEXTBOOK_LIB_PATH = os.path.join(HERE, "libtfhe_hip_textbook.so")

SYMBOLS = [
    "tfhe_hip_variant", "tfhe_hip_create", "tfhe_hip_destroy", "tfhe_hip_last_error", "tfhe_hip_key_transport", "tfhe_hip_set_stream", "tfhe_hip_set_blind_rotate_mode",
    "tfhe_hip_circuit_add", "tfhe_hip_circuit_add_dev",
    "tfhe_hip_synchronize", "tfhe_hip_load_keys", "tfhe_hip_key_blob_sizes",
    "tfhe_hip_export_keys_dev", "tfhe_hip_import_keys_dev", "tfhe_hip_bootstrap",
    "tfhe_hip_bootstrap_dev", "tfhe_hip_bootstrap_noks", "tfhe_hip_bootstrap_noks_dev",
    "tfhe_hip_keyswitch", "tfhe_hip_keyswitch_dev", "tfhe_hip_gate", "tfhe_hip_gate_dev",
    "tfhe_hip_polymul", "tfhe_hip_polymul_dev", "tfhe_hip_extprod", "tfhe_hip_blind_rotate",
    "tfhe_hip_profile_enable", "tfhe_hip_profile_read",
    "tfhe_hip_create_multi", "tfhe_hip_key_file_size", "tfhe_hip_key_file_pack", "tfhe_hip_load_keys_blob",
    "tfhe_hip_not_dev", "tfhe_hip_constant_dev",
    "tfhe_hip_samples_alloc", "tfhe_hip_samples_free", "tfhe_hip_samples_count", "tfhe_hip_samples_upload",
    "tfhe_hip_samples_download", "tfhe_hip_samples_gate", "tfhe_hip_samples_bootstrap",
    "tfhe_hip_samples_not", "tfhe_hip_samples_constant",
    "tfhe_hip_polymul_margin", "tfhe_hip_polymul_margin_dev", "tfhe_hip_extprod_margin",
    "tfhe_hip_extprod_dev", "tfhe_hip_blind_rotate_dev", "tfhe_hip_gate_dev_noalloc",
    "tfhe_hip_generate_keys", "tfhe_hip_keygen_stream",
    "tfhe_hip_circuit_compare", "tfhe_hip_circuit_compare_dev", "tfhe_hip_circuit_eq", "tfhe_hip_circuit_eq_dev",
]
NO_INT_RESULT = ("tfhe_hip_variant", "tfhe_hip_destroy", "tfhe_hip_last_error", "tfhe_hip_key_transport", "tfhe_hip_samples_free", "tfhe_hip_samples_count")


class Params(C.Structure):
    """tfhe_hip_params"""
    _fields_ = [(f, C.c_int32) for f in
                ("n", "N", "k", "l", "bg_bit", "ks_t", "ks_base_bit", "n_ks")]


def load(variant="reference"):
    path = {"reference": LIB_PATH, "textbook": TEXTBOOK_LIB_PATH}[variant]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP library must be built (there is no CPU fallback)")
    try:
        # If torch is going to be used in this process it must own the HIP runtime.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    vp, i32p, f64p, sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t
    PP = C.POINTER(Params)
    lib.tfhe_hip_variant.argtypes = []
    lib.tfhe_hip_variant.restype = C.c_char_p
    lib.tfhe_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.tfhe_hip_destroy.argtypes = [vp]
    lib.tfhe_hip_destroy.restype = None
    lib.tfhe_hip_last_error.argtypes = [vp]
    lib.tfhe_hip_last_error.restype = C.c_char_p
    lib.tfhe_hip_key_transport.argtypes = [vp]
    lib.tfhe_hip_key_transport.restype = C.c_char_p
    lib.tfhe_hip_set_stream.argtypes = [vp, vp]
    lib.tfhe_hip_synchronize.argtypes = [vp]
    lib.tfhe_hip_set_blind_rotate_mode.argtypes = [vp, C.c_int]
    lib.tfhe_hip_load_keys.argtypes = [vp, PP, i32p, i32p, i32p, f64p]
    lib.tfhe_hip_key_blob_sizes.argtypes = [PP, C.POINTER(sz), C.POINTER(sz)]
    lib.tfhe_hip_export_keys_dev.argtypes = [vp, vp, vp]
    lib.tfhe_hip_import_keys_dev.argtypes = [vp, PP, vp, vp]
    for name in ("tfhe_hip_bootstrap", "tfhe_hip_bootstrap_dev"):
        getattr(lib, name).argtypes = [vp, C.c_int32, i32p, i32p, i32p, i32p, f64p, sz]
    for name in ("tfhe_hip_bootstrap_noks", "tfhe_hip_bootstrap_noks_dev"):
        getattr(lib, name).argtypes = [vp, C.c_int32, i32p, i32p, i32p, i32p, sz]
    for name in ("tfhe_hip_keyswitch", "tfhe_hip_keyswitch_dev"):
        getattr(lib, name).argtypes = [vp, i32p, i32p, i32p, i32p, f64p, sz]
    for name in ("tfhe_hip_gate", "tfhe_hip_gate_dev", "tfhe_hip_gate_dev_noalloc"):
        getattr(lib, name).argtypes = [vp, C.c_int, i32p, i32p, i32p, i32p, i32p, i32p, i32p, i32p,
                                       f64p, sz]
    for name in ("tfhe_hip_polymul", "tfhe_hip_polymul_dev"):
        getattr(lib, name).argtypes = [vp, i32p, i32p, i32p, C.c_int, sz]
    for name in ("tfhe_hip_circuit_add", "tfhe_hip_circuit_add_dev"):
        getattr(lib, name).argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p, i32p, i32p, f64p, f64p, sz, C.c_int]
    for name in ("tfhe_hip_circuit_compare", "tfhe_hip_circuit_compare_dev", "tfhe_hip_circuit_eq",
                 "tfhe_hip_circuit_eq_dev"):
        getattr(lib, name).argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p, f64p, sz, C.c_int]
    lib.tfhe_hip_profile_enable.argtypes = [vp, C.c_int]
    lib.tfhe_hip_profile_read.argtypes = [vp, C.POINTER(C.c_double * 3), C.POINTER(C.c_int * 3)]
    lib.tfhe_hip_extprod.argtypes = [vp, i32p, C.c_int, i32p, sz]
    lib.tfhe_hip_extprod_margin.argtypes = [vp, i32p, C.c_int, i32p, f64p, sz]
    lib.tfhe_hip_extprod_dev.argtypes = [vp, i32p, C.c_int, i32p, f64p, sz]
    for name in ("tfhe_hip_blind_rotate", "tfhe_hip_blind_rotate_dev"):
        getattr(lib, name).argtypes = [vp, i32p, i32p, sz]
    for name in ("tfhe_hip_polymul_margin", "tfhe_hip_polymul_margin_dev"):
        getattr(lib, name).argtypes = [vp, i32p, i32p, f64p, sz]
    lib.tfhe_hip_create_multi.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    lib.tfhe_hip_key_file_size.argtypes = [PP, C.c_int, C.POINTER(sz)]
    lib.tfhe_hip_key_file_pack.argtypes = [PP, i32p, i32p, i32p, f64p, vp, sz]
    lib.tfhe_hip_load_keys_blob.argtypes = [vp, vp, sz]
    lib.tfhe_hip_generate_keys.argtypes = [vp, PP, C.c_double, C.c_double, vp, i32p, i32p, i32p, i32p, i32p, f64p]
    lib.tfhe_hip_keygen_stream.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, sz]
    lib.tfhe_hip_not_dev.argtypes = [vp, i32p, i32p, f64p, i32p, i32p, f64p, sz]
    lib.tfhe_hip_constant_dev.argtypes = [vp, C.c_int, i32p, i32p, f64p, sz]
    lib.tfhe_hip_samples_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.tfhe_hip_samples_free.argtypes = [vp]
    lib.tfhe_hip_samples_free.restype = None
    lib.tfhe_hip_samples_count.argtypes = [vp]
    lib.tfhe_hip_samples_count.restype = sz
    lib.tfhe_hip_samples_upload.argtypes = [vp, vp, i32p, i32p, f64p]
    lib.tfhe_hip_samples_download.argtypes = [vp, vp, i32p, i32p, f64p]
    lib.tfhe_hip_samples_gate.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.tfhe_hip_samples_bootstrap.argtypes = [vp, C.c_int32, vp, vp]
    lib.tfhe_hip_samples_not.argtypes = [vp, vp, vp]
    lib.tfhe_hip_samples_constant.argtypes = [vp, C.c_int, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in NO_INT_RESULT:
            fn.restype = C.c_int
    if "TFHE_HIP_LIB" not in os.environ and lib.tfhe_hip_variant().decode() != variant:
        raise ImportError(f"{path} is not the {variant} build")
    return lib
