"""Thin Python handle over the C ABI (include/tfhe_hip.h).

``Engine`` owns one ``tfhe_hip_ctx`` on one GPU (``Engine(0)``) or on a list of
GPUs (``Engine([0, 1, 2, 3])``, tfhe_hip_create_multi).  Methods taking numpy
arrays use the host-pointer entry points; the ``*_dev`` methods take torch CUDA
tensors, run on torch's current stream and do not synchronise (single-GPU
engines only).  Every array is checked against the batch size and the loaded
parameters before its pointer crosses the boundary.
"""
import ctypes as C

import numpy as np

from . import _lib

GATES = ["nand", "or", "and", "xor", "xnor", "nor", "andny", "andyn", "orny", "oryn", "mux"]
GATE_ID = {g: i for i, g in enumerate(GATES)}


class TfheHipError(RuntimeError):
    pass


def make_params(n, N=1024, k=1, l=2, bg_bit=10, ks_t=8, ks_base_bit=2, n_ks=None):
    return _lib.Params(n, N, k, l, bg_bit, ks_t, ks_base_bit, n if n_ks is None else n_ks)


def _np(a, dtype=np.int32):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _tp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _want(arr, shape, what):
    if tuple(arr.shape) != tuple(shape):
        raise ValueError(f"{what}: expected shape {tuple(shape)}, got {tuple(arr.shape)}")
    return arr


class Samples:
    """Device-resident batch of LWE samples (tfhe_hip_samples)."""

    def __init__(self, engine, B):
        self.engine = engine
        self.handle = C.c_void_p()
        engine._check(engine.lib.tfhe_hip_samples_alloc(engine.ctx, B, C.byref(self.handle)))
        self.B = B

    def free(self):
        if self.handle:
            self.engine.lib.tfhe_hip_samples_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        # safe after Engine.close(): destroying a context releases the device memory of its
        # live handles, the free that follows only deletes the handle
        try:
            self.free()
        except Exception:
            pass

    def upload(self, a, b, var=None):
        e = self.engine
        a = _want(_np(a), (self.B, e.params.n), "a")
        b = _want(_np(b), (self.B,), "b")
        var = None if var is None else _want(_np(var, np.float64), (self.B,), "var")
        e._check(e.lib.tfhe_hip_samples_upload(e.ctx, self.handle, _p(a), _p(b), _p(var)))
        return self

    def download(self, want_var=False):
        e = self.engine
        a = np.empty((self.B, e.params.n), np.int32)
        b = np.empty(self.B, np.int32)
        var = np.empty(self.B, np.float64) if want_var else None
        e._check(e.lib.tfhe_hip_samples_download(e.ctx, self.handle, _p(a), _p(b), _p(var)))
        return (a, b, var) if want_var else (a, b)


class Engine:
    def __init__(self, device=0, variant="reference"):
        """variant: "reference" (libtfhe_hip.so, the crate's results) or "textbook"
        (libtfhe_hip_textbook.so, the corrected scheme; same entry points)."""
        self.variant = variant
        self.lib = _lib.load(variant)
        self.ctx = C.c_void_p()
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*device)
            rc = self.lib.tfhe_hip_create_multi(C.byref(self.ctx), devs, len(device))
        else:
            rc = self.lib.tfhe_hip_create(C.byref(self.ctx), device)
        if rc != 0:
            raise TfheHipError(f"tfhe_hip_create failed ({rc}): no usable GPU {device}")
        self.device = device
        self.params = None

    def close(self):
        if self.ctx:
            self.lib.tfhe_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise TfheHipError(f"{rc}: {self.lib.tfhe_hip_last_error(self.ctx).decode()}")

    def last_message(self):
        return self.lib.tfhe_hip_last_error(self.ctx).decode()

    def key_transport(self):
        """How a multi-GPU context's keys reached its other devices: rccl / peer / local / none."""
        return self.lib.tfhe_hip_key_transport(self.ctx).decode()

    # ---- keys -----------------------------------------------------------------
    @staticmethod
    def _key_arrays(params, bk, ks_a, ks_b, ks_var):
        base = 1 << params.ks_base_bit
        bk = _np(bk)
        ks_a, ks_b = _np(ks_a), _np(ks_b)
        ks_var = None if ks_var is None else _np(ks_var, np.float64)
        rows = params.n_ks * params.ks_t * base
        if bk.size != params.n * params.l * (params.k + 1) ** 2 * params.N:
            raise ValueError("bk does not match the parameters")
        if ks_a.size != rows * params.n or ks_b.size != rows:
            raise ValueError("ks_a / ks_b do not match the parameters")
        if ks_var is not None and ks_var.size != rows:
            raise ValueError("ks_var does not match the parameters")
        return bk, ks_a, ks_b, ks_var

    def load_keys(self, params, bk, ks_a, ks_b, ks_var=None):
        bk, ks_a, ks_b, ks_var = self._key_arrays(params, bk, ks_a, ks_b, ks_var)
        self._check(self.lib.tfhe_hip_load_keys(self.ctx, C.byref(params), _p(bk), _p(ks_a),
                                                _p(ks_b), _p(ks_var)))
        self.params = params

    def generate_keys(self, params, ks_stdev, bk_stdev, seed, export=True):
        """generate_keys on the GPU (tfhe_hip_generate_keys).  seed: 32 bytes.  Returns a dict
        with the secret keys (lwe_key [n], tlwe_key [k+1][N]) and, with export, the cloud key
        arrays bk, ks_a, ks_b, ks_var in the layouts of load_keys.  The engine holds the keys."""
        seed = np.frombuffer(bytes(seed), dtype=np.uint8)
        if seed.size != 32:
            raise ValueError("seed must be 32 bytes")
        seed = np.ascontiguousarray(seed)
        base = 1 << params.ks_base_bit
        out = {"lwe_key": np.empty(params.n, np.int32),
               "tlwe_key": np.empty((params.k + 1, params.N), np.int32)}
        if export:
            out["bk"] = np.empty((params.n, params.l, params.k + 1, params.k + 1, params.N), np.int32)
            out["ks_a"] = np.empty((params.n_ks, params.ks_t, base, params.n), np.int32)
            out["ks_b"] = np.empty((params.n_ks, params.ks_t, base), np.int32)
            out["ks_var"] = np.empty((params.n_ks, params.ks_t, base), np.float64)
        self._check(self.lib.tfhe_hip_generate_keys(
            self.ctx, C.byref(params), float(ks_stdev), float(bk_stdev), _p(seed), _p(out["lwe_key"]),
            _p(out["tlwe_key"]), _p(out.get("bk")), _p(out.get("ks_a")), _p(out.get("ks_b")),
            _p(out.get("ks_var"))))
        self.params = params
        return out

    def keygen_stream(self, seed, purpose, stream, first_block, nblocks):
        """Raw ChaCha20 keystream words [nblocks][16] behind generate_keys (test hook)."""
        seed = np.ascontiguousarray(np.frombuffer(bytes(seed), dtype=np.uint8))
        if seed.size != 32:
            raise ValueError("seed must be 32 bytes")
        out = np.empty((nblocks, 16), np.uint32)
        self._check(self.lib.tfhe_hip_keygen_stream(self.ctx, _p(seed), purpose, stream, first_block,
                                                    _p(out), nblocks))
        return out

    def pack_key_file(self, params, bk, ks_a, ks_b, ks_var=None):
        """The flat key image (bytes) that load_key_file / tfhe_hip_load_keys_blob read."""
        bk, ks_a, ks_b, ks_var = self._key_arrays(params, bk, ks_a, ks_b, ks_var)
        size = C.c_size_t()
        if self.lib.tfhe_hip_key_file_size(C.byref(params), int(ks_var is not None), C.byref(size)) != 0:
            raise TfheHipError("unsupported parameters")
        out = np.empty(size.value, np.uint8)
        if self.lib.tfhe_hip_key_file_pack(C.byref(params), _p(bk), _p(ks_a), _p(ks_b), _p(ks_var),
                                           _p(out), out.size) != 0:
            raise TfheHipError("tfhe_hip_key_file_pack failed")
        return out

    def load_key_file(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        self._check(self.lib.tfhe_hip_load_keys_blob(self.ctx, _p(image), image.size))
        f = np.frombuffer(image[16:48].tobytes(), dtype=np.int32)
        self.params = _lib.Params(*[int(x) for x in f])

    def key_blob_sizes(self, params=None):
        params = params or self.params
        a, b = C.c_size_t(), C.c_size_t()
        rc = self.lib.tfhe_hip_key_blob_sizes(C.byref(params), C.byref(a), C.byref(b))
        if rc != 0:
            raise TfheHipError("unsupported parameters")
        return a.value, b.value

    def export_keys_dev(self, bk_blob, ks_blob):
        bk_bytes, ks_bytes = self.key_blob_sizes()
        if bk_blob.numel() * bk_blob.element_size() < bk_bytes or ks_blob.numel() * ks_blob.element_size() < ks_bytes:
            raise ValueError("key blob buffers are too small")
        self._check(self.lib.tfhe_hip_export_keys_dev(self.ctx, _tp(bk_blob), _tp(ks_blob)))

    def import_keys_dev(self, params, bk_blob, ks_blob):
        bk_bytes, ks_bytes = self.key_blob_sizes(params)
        if bk_blob.numel() * bk_blob.element_size() < bk_bytes or ks_blob.numel() * ks_blob.element_size() < ks_bytes:
            raise ValueError("key blob buffers are too small")
        self._check(self.lib.tfhe_hip_import_keys_dev(self.ctx, C.byref(params), _tp(bk_blob),
                                                      _tp(ks_blob)))
        self.params = params

    def use_torch_stream(self):
        import torch
        self._check(self.lib.tfhe_hip_set_stream(
            self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    BR_MODES = {"auto": 0, "batch": 1, "split": 2, "batch64": 3}

    def set_blind_rotate_mode(self, mode):
        """'batch' / 'batch64' = one ciphertext per wavefront with one / two wavefronts per SIMD,
        'split' = one per workgroup, 'auto' by batch size and parameter set."""
        self._check(self.lib.tfhe_hip_set_blind_rotate_mode(self.ctx, self.BR_MODES[mode]))

    def profile_enable(self, on=True):
        self._check(self.lib.tfhe_hip_profile_enable(self.ctx, int(on)))

    def profile_read(self):
        """{kernel: (summed ms, launches)} since the last read; waits for the stream."""
        ms, cnt = (C.c_double * 3)(), (C.c_int * 3)()
        self._check(self.lib.tfhe_hip_profile_read(self.ctx, C.byref(ms), C.byref(cnt)))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(("prologue", "blind_rotate", "keyswitch"))}

    def synchronize(self):
        self._check(self.lib.tfhe_hip_synchronize(self.ctx))

    # ---- host-pointer path ------------------------------------------------------
    def _sample(self, pair, B, width, what):
        a, b = _np(pair[0]), _np(pair[1])
        if B is None:
            B = b.size
        return _want(a, (B, width), what + " mask"), _want(b.reshape(-1), (B,), what + " body"), B

    def bootstrap(self, mu, a, b, want_var=False):
        a, b, B = self._sample((a, b), None, self.params.n, "x")
        out_a = np.empty((B, self.params.n), np.int32)
        out_b = np.empty(B, np.int32)
        var = np.empty(B, np.float64) if want_var else None
        self._check(self.lib.tfhe_hip_bootstrap(self.ctx, mu, _p(a), _p(b), _p(out_a), _p(out_b),
                                                _p(var), B))
        return (out_a, out_b, var) if want_var else (out_a, out_b)

    def bootstrap_noks(self, mu, a, b):
        a, b, B = self._sample((a, b), None, self.params.n, "x")
        out_a = np.empty((B, self.params.N * self.params.k), np.int32)
        out_b = np.empty(B, np.int32)
        self._check(self.lib.tfhe_hip_bootstrap_noks(self.ctx, mu, _p(a), _p(b), _p(out_a),
                                                     _p(out_b), B))
        return out_a, out_b

    def keyswitch(self, a, b, want_var=False):
        a, b, B = self._sample((a, b), None, self.params.N * self.params.k, "x")
        out_a = np.empty((B, self.params.n), np.int32)
        out_b = np.empty(B, np.int32)
        var = np.empty(B, np.float64) if want_var else None
        self._check(self.lib.tfhe_hip_keyswitch(self.ctx, _p(a), _p(b), _p(out_a), _p(out_b),
                                                _p(var), B))
        return (out_a, out_b, var) if want_var else (out_a, out_b)

    def gate(self, gate, ca, cb, cc=None, want_var=False):
        """ca, cb, cc: (a [B][n], b [B]) pairs."""
        n = self.params.n
        ca_a, ca_b, B = self._sample(ca, None, n, "ca")
        cb_a, cb_b, _ = self._sample(cb, B, n, "cb")
        cc_a = cc_b = None
        if cc is not None:
            cc_a, cc_b, _ = self._sample(cc, B, n, "cc")
        out_a = np.empty((B, n), np.int32)
        out_b = np.empty(B, np.int32)
        var = np.empty(B, np.float64) if want_var else None
        self._check(self.lib.tfhe_hip_gate(self.ctx, GATE_ID[gate], _p(ca_a), _p(ca_b), _p(cb_a),
                                           _p(cb_b), _p(cc_a), _p(cc_b), _p(out_a), _p(out_b),
                                           _p(var), B))
        return (out_a, out_b, var) if want_var else (out_a, out_b)

    def circuit_add(self, a, b, want_var=False):
        """circuits::add on many instances.  a, b: (masks [bits][count][n], bodies [bits][count]),
        bit 0 first.  Returns (sum, carry[, sum_var, carry_var]) with sum = (masks, bodies) in the
        same layout and carry = ([count][n], [count])."""
        a_a, a_b, b_a, b_b = _np(a[0]), _np(a[1]), _np(b[0]), _np(b[1])
        if a_b.ndim != 2:
            raise ValueError("bodies must be [bits][count]")
        bits, count = a_b.shape
        n = self.params.n
        _want(a_a, (bits, count, n), "a masks")
        _want(b_a, (bits, count, n), "b masks")
        _want(b_b, (bits, count), "b bodies")
        sum_a, sum_b = np.empty((bits, count, n), np.int32), np.empty((bits, count), np.int32)
        car_a, car_b = np.empty((count, n), np.int32), np.empty(count, np.int32)
        sv = np.empty((bits, count), np.float64) if want_var else None
        cv = np.empty(count, np.float64) if want_var else None
        self._check(self.lib.tfhe_hip_circuit_add(self.ctx, _p(a_a), _p(a_b), _p(b_a), _p(b_b),
                                                  _p(sum_a), _p(sum_b), _p(car_a), _p(car_b), _p(sv),
                                                  _p(cv), count, bits))
        if want_var:
            return (sum_a, sum_b), (car_a, car_b), sv, cv
        return (sum_a, sum_b), (car_a, car_b)

    def _circuit_reduce(self, fn, a, b, want_var):
        a_a, a_b, b_a, b_b = _np(a[0]), _np(a[1]), _np(b[0]), _np(b[1])
        if a_b.ndim != 2:
            raise ValueError("bodies must be [bits][count]")
        bits, count = a_b.shape
        n = self.params.n
        _want(a_a, (bits, count, n), "a masks")
        _want(b_a, (bits, count, n), "b masks")
        _want(b_b, (bits, count), "b bodies")
        out_a, out_b = np.empty((count, n), np.int32), np.empty(count, np.int32)
        var = np.empty(count, np.float64) if want_var else None
        self._check(fn(self.ctx, _p(a_a), _p(a_b), _p(b_a), _p(b_b), _p(out_a), _p(out_b), _p(var),
                       count, bits))
        return (out_a, out_b, var) if want_var else (out_a, out_b)

    def circuit_compare(self, a, b, want_var=False):
        """circuits::compare (a <= b) on many instances; operands as for circuit_add."""
        return self._circuit_reduce(self.lib.tfhe_hip_circuit_compare, a, b, want_var)

    def circuit_eq(self, a, b, want_var=False):
        """circuits::eq on many instances; operands as for circuit_add."""
        return self._circuit_reduce(self.lib.tfhe_hip_circuit_eq, a, b, want_var)

    def polymul(self, d, s, accumulate_into=None):
        d, s = _np(d), _np(s)
        if d.ndim != 2 or d.shape[1] != 1024:
            raise ValueError("d must be [B][1024]")
        B = d.shape[0]
        _want(s, (B, 1024), "s")
        if accumulate_into is None:
            out = np.empty((B, 2 * d.shape[1] - 1), np.int32)
            acc = 0
        else:
            out = _want(np.array(accumulate_into, dtype=np.int32, order="C"), (B, 1024), "accumulate_into")
            acc = 1
        self._check(self.lib.tfhe_hip_polymul(self.ctx, _p(d), _p(s), _p(out), acc, B))
        return out

    def polymul_margin(self, d, s):
        """[B][2]: largest |x - rint(x)| before rounding, over all coefficients and over those
        whose rounding decides the result, measured in the product kernel itself."""
        d, s = _np(d), _np(s)
        if d.ndim != 2 or d.shape[1] != 1024:
            raise ValueError("d must be [B][1024]")
        B = d.shape[0]
        _want(s, (B, 1024), "s")
        m = np.empty((B, 2), np.float64)
        self._check(self.lib.tfhe_hip_polymul_margin(self.ctx, _p(d), _p(s), _p(m), B))
        return m

    def extprod(self, t, bk_index, want_margin=False):
        t = _np(t)
        if t.ndim != 3 or t.shape[1:] != (3, 1024):
            raise ValueError("t must be [B][3][1024]")
        out = np.empty_like(t)
        if not want_margin:
            self._check(self.lib.tfhe_hip_extprod(self.ctx, _p(t), bk_index, _p(out), t.shape[0]))
            return out
        m = np.empty((t.shape[0], 2), np.float64)
        self._check(self.lib.tfhe_hip_extprod_margin(self.ctx, _p(t), bk_index, _p(out), _p(m), t.shape[0]))
        return out, m

    def blind_rotate(self, acc, bara):
        acc = np.array(acc, dtype=np.int32, order="C")
        if acc.ndim != 3 or acc.shape[1:] != (3, 1024):
            raise ValueError("acc must be [B][3][1024]")
        bara = _want(_np(bara), (acc.shape[0], self.params.n), "bara")
        self._check(self.lib.tfhe_hip_blind_rotate(self.ctx, _p(acc), _p(bara), acc.shape[0]))
        return acc

    # ---- device-resident sample handles -------------------------------------------
    def samples(self, B):
        return Samples(self, B)

    def samples_from(self, a, b, var=None):
        b = _np(b).reshape(-1)
        return Samples(self, b.size).upload(a, b, var)

    def samples_gate(self, gate, ca, cb, out, cc=None):
        self._check(self.lib.tfhe_hip_samples_gate(self.ctx, GATE_ID[gate], ca.handle, cb.handle,
                                                   cc.handle if cc is not None else None, out.handle))
        return out

    def samples_bootstrap(self, mu, x, out):
        self._check(self.lib.tfhe_hip_samples_bootstrap(self.ctx, mu, x.handle, out.handle))
        return out

    def samples_not(self, x, out):
        self._check(self.lib.tfhe_hip_samples_not(self.ctx, x.handle, out.handle))
        return out

    def samples_constant(self, value, out):
        self._check(self.lib.tfhe_hip_samples_constant(self.ctx, int(bool(value)), out.handle))
        return out

    # ---- device-pointer path (torch tensors on this engine's GPU) -----------------
    def _dev_sample(self, a, b, B, width, what):
        if a.numel() != B * width or b.numel() != B:
            raise ValueError(f"{what}: expected [{B}][{width}] and [{B}]")

    def gate_dev(self, gate, ca_a, ca_b, cb_a, cb_b, out_a, out_b, cc_a=None, cc_b=None, var=None):
        B, n = ca_b.numel(), self.params.n
        self._dev_sample(ca_a, ca_b, B, n, "ca")
        self._dev_sample(cb_a, cb_b, B, n, "cb")
        self._dev_sample(out_a, out_b, B, n, "out")
        if cc_a is not None:
            self._dev_sample(cc_a, cc_b, B, n, "cc")
        self._check(self.lib.tfhe_hip_gate_dev(self.ctx, GATE_ID[gate], _tp(ca_a), _tp(ca_b),
                                               _tp(cb_a), _tp(cb_b), _tp(cc_a), _tp(cc_b),
                                               _tp(out_a), _tp(out_b), _tp(var), B))

    def gate_dev_noalloc(self, gate, ca_a, ca_b, cb_a, cb_b, out_a, out_b, cc_a=None, cc_b=None, var=None):
        """gate_dev through the entry point that allocates nothing (fixed scratch, bounded batch)."""
        B, n = ca_b.numel(), self.params.n
        self._dev_sample(ca_a, ca_b, B, n, "ca")
        self._dev_sample(cb_a, cb_b, B, n, "cb")
        self._dev_sample(out_a, out_b, B, n, "out")
        if cc_a is not None:
            self._dev_sample(cc_a, cc_b, B, n, "cc")
        self._check(self.lib.tfhe_hip_gate_dev_noalloc(self.ctx, GATE_ID[gate], _tp(ca_a), _tp(ca_b),
                                                       _tp(cb_a), _tp(cb_b), _tp(cc_a), _tp(cc_b),
                                                       _tp(out_a), _tp(out_b), _tp(var), B))

    def circuit_add_dev(self, a_a, a_b, b_a, b_b, sum_a, sum_b, carry_a, carry_b, sum_var=None,
                        carry_var=None):
        bits, count = a_b.shape
        n = self.params.n
        for t, w in ((a_a, n), (b_a, n), (sum_a, n), (a_b, 1), (b_b, 1), (sum_b, 1)):
            if t.numel() != bits * count * w:
                raise ValueError("operand shapes differ")
        self._dev_sample(carry_a, carry_b, count, n, "carry")
        self._check(self.lib.tfhe_hip_circuit_add_dev(
            self.ctx, _tp(a_a), _tp(a_b), _tp(b_a), _tp(b_b), _tp(sum_a), _tp(sum_b), _tp(carry_a),
            _tp(carry_b), _tp(sum_var), _tp(carry_var), count, bits))

    def bootstrap_dev(self, mu, a, b, out_a, out_b, var=None):
        B, n = b.numel(), self.params.n
        self._dev_sample(a, b, B, n, "x")
        self._dev_sample(out_a, out_b, B, n, "out")
        self._check(self.lib.tfhe_hip_bootstrap_dev(self.ctx, mu, _tp(a), _tp(b), _tp(out_a),
                                                    _tp(out_b), _tp(var), B))

    def polymul_dev(self, d, s, out, accumulate=False):
        B = d.shape[0]
        if d.numel() != B * 1024 or s.numel() != B * 1024 or out.numel() != B * (1024 if accumulate else 2047):
            raise ValueError("polynomial shapes differ")
        self._check(self.lib.tfhe_hip_polymul_dev(self.ctx, _tp(d), _tp(s), _tp(out),
                                                  int(accumulate), B))
