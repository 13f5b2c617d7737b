"""Thin Python handle over the C ABI (include/tfhe_hip.h).

``Engine`` owns one ``tfhe_hip_ctx`` on one GPU.  Methods taking numpy arrays use
the host-pointer entry points; the ``*_dev`` methods take torch CUDA tensors,
run on torch's current stream and do not synchronise.
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


class Engine:
    def __init__(self, device=0):
        self.lib = _lib.load()
        self.ctx = C.c_void_p()
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

    # ---- keys -----------------------------------------------------------------
    def load_keys(self, params, bk, ks_a, ks_b, ks_var=None):
        bk, ks_a, ks_b = _np(bk), _np(ks_a), _np(ks_b)
        ks_var = None if ks_var is None else _np(ks_var, np.float64)
        assert bk.size == params.n * params.l * (params.k + 1) ** 2 * params.N
        assert ks_a.size == params.n_ks * params.ks_t * (1 << params.ks_base_bit) * params.n
        self._check(self.lib.tfhe_hip_load_keys(self.ctx, C.byref(params), _p(bk), _p(ks_a),
                                                _p(ks_b), _p(ks_var)))
        self.params = params

    def key_blob_sizes(self, params=None):
        params = params or self.params
        a, b = C.c_size_t(), C.c_size_t()
        rc = self.lib.tfhe_hip_key_blob_sizes(C.byref(params), C.byref(a), C.byref(b))
        if rc != 0:
            raise TfheHipError("unsupported parameters")
        return a.value, b.value

    def export_keys_dev(self, bk_blob, ks_blob):
        self._check(self.lib.tfhe_hip_export_keys_dev(self.ctx, _tp(bk_blob), _tp(ks_blob)))

    def import_keys_dev(self, params, bk_blob, ks_blob):
        self._check(self.lib.tfhe_hip_import_keys_dev(self.ctx, C.byref(params), _tp(bk_blob),
                                                      _tp(ks_blob)))
        self.params = params

    def use_torch_stream(self):
        import torch
        self._check(self.lib.tfhe_hip_set_stream(
            self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    BR_MODES = {"auto": 0, "batch": 1, "split": 2}

    def set_blind_rotate_mode(self, mode):
        """'batch' = one ciphertext per wavefront, 'split' = one per workgroup, 'auto' by batch size."""
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
    def bootstrap(self, mu, a, b, want_var=False):
        a, b = _np(a), _np(b)
        B = b.size
        out_a = np.empty((B, self.params.n), np.int32)
        out_b = np.empty(B, np.int32)
        var = np.empty(B, np.float64) if want_var else None
        self._check(self.lib.tfhe_hip_bootstrap(self.ctx, mu, _p(a), _p(b), _p(out_a), _p(out_b),
                                                _p(var), B))
        return (out_a, out_b, var) if want_var else (out_a, out_b)

    def bootstrap_noks(self, mu, a, b):
        a, b = _np(a), _np(b)
        B = b.size
        out_a = np.empty((B, self.params.N * self.params.k), np.int32)
        out_b = np.empty(B, np.int32)
        self._check(self.lib.tfhe_hip_bootstrap_noks(self.ctx, mu, _p(a), _p(b), _p(out_a),
                                                     _p(out_b), B))
        return out_a, out_b

    def keyswitch(self, a, b, want_var=False):
        a, b = _np(a), _np(b)
        B = b.size
        out_a = np.empty((B, self.params.n), np.int32)
        out_b = np.empty(B, np.int32)
        var = np.empty(B, np.float64) if want_var else None
        self._check(self.lib.tfhe_hip_keyswitch(self.ctx, _p(a), _p(b), _p(out_a), _p(out_b),
                                                _p(var), B))
        return (out_a, out_b, var) if want_var else (out_a, out_b)

    def gate(self, gate, ca, cb, cc=None, want_var=False):
        """ca, cb, cc: (a [B][n], b [B]) pairs."""
        ca_a, ca_b, cb_a, cb_b = _np(ca[0]), _np(ca[1]), _np(cb[0]), _np(cb[1])
        cc_a = None if cc is None else _np(cc[0])
        cc_b = None if cc is None else _np(cc[1])
        B = ca_b.size
        out_a = np.empty((B, self.params.n), np.int32)
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
        bits, count = a_b.shape
        n = self.params.n
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

    def polymul(self, d, s, accumulate_into=None):
        d, s = _np(d), _np(s)
        B = d.shape[0]
        if accumulate_into is None:
            out = np.empty((B, 2 * d.shape[1] - 1), np.int32)
            acc = 0
        else:
            out = np.array(accumulate_into, dtype=np.int32, order="C")
            acc = 1
        self._check(self.lib.tfhe_hip_polymul(self.ctx, _p(d), _p(s), _p(out), acc, B))
        return out

    def extprod(self, t, bk_index):
        t = _np(t)
        out = np.empty_like(t)
        self._check(self.lib.tfhe_hip_extprod(self.ctx, _p(t), bk_index, _p(out), t.shape[0]))
        return out

    def blind_rotate(self, acc, bara):
        acc = np.array(acc, dtype=np.int32, order="C")
        bara = _np(bara)
        self._check(self.lib.tfhe_hip_blind_rotate(self.ctx, _p(acc), _p(bara), acc.shape[0]))
        return acc

    # ---- device-pointer path (torch tensors on this engine's GPU) -----------------
    def gate_dev(self, gate, ca_a, ca_b, cb_a, cb_b, out_a, out_b, cc_a=None, cc_b=None, var=None):
        self._check(self.lib.tfhe_hip_gate_dev(self.ctx, GATE_ID[gate], _tp(ca_a), _tp(ca_b),
                                               _tp(cb_a), _tp(cb_b), _tp(cc_a), _tp(cc_b),
                                               _tp(out_a), _tp(out_b), _tp(var), ca_b.numel()))

    def circuit_add_dev(self, a_a, a_b, b_a, b_b, sum_a, sum_b, carry_a, carry_b, sum_var=None,
                        carry_var=None):
        bits, count = a_b.shape
        self._check(self.lib.tfhe_hip_circuit_add_dev(
            self.ctx, _tp(a_a), _tp(a_b), _tp(b_a), _tp(b_b), _tp(sum_a), _tp(sum_b), _tp(carry_a),
            _tp(carry_b), _tp(sum_var), _tp(carry_var), count, bits))

    def bootstrap_dev(self, mu, a, b, out_a, out_b, var=None):
        self._check(self.lib.tfhe_hip_bootstrap_dev(self.ctx, mu, _tp(a), _tp(b), _tp(out_a),
                                                    _tp(out_b), _tp(var), b.numel()))

    def polymul_dev(self, d, s, out, accumulate=False):
        self._check(self.lib.tfhe_hip_polymul_dev(self.ctx, _tp(d), _tp(s), _tp(out),
                                                  int(accumulate), d.shape[0]))
