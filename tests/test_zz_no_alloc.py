"""tfhe_hip_gate_dev_noalloc, the entry point that allocates nothing (its scratch is a fixed
array inside the library), against tfhe_hip_gate_dev on the same device buffers: two calls in
a row, the second with a smaller batch, NAND and mux, results and variances bit for bit; a
batch beyond the array is refused with a message.  Called through the C ABI."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu


def test_gate_dev_noalloc_equals_gate_dev():
    import torch
    import tfhe_amd
    dev = torch.device("cuda", 0)
    P = orc.params(80)
    P.n = P.n_ks = 40                                  # full-size polynomials, short LWE dimension
    K = orc.generate_synthetic_keys(P, 3100)
    e = tfhe_amd.Engine(0)
    try:
        e.load_keys(tfhe_amd.make_params(P.n, n_ks=P.n_ks), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])

        def tens(seed, B):
            return (torch.from_numpy(orc.uniform_i32(seed, (B, P.n))).to(dev),
                    torch.from_numpy(orc.uniform_i32(seed + 1, (B,))).to(dev))

        def outs(B):
            return (torch.empty((B, P.n), dtype=torch.int32, device=dev),
                    torch.empty((B,), dtype=torch.int32, device=dev),
                    torch.empty((B,), dtype=torch.float64, device=dev))

        for call, (B, gate) in enumerate(((300, "nand"), (37, "mux"), (5, "nand"))):
            ca, cb, cc = tens(3200 + 10 * call, B), tens(3300 + 10 * call, B), tens(3400 + 10 * call, B)
            third = cc if gate == "mux" else (None, None)
            want, got = outs(B), outs(B)
            for o in got:
                o.fill_(-7)
            e.gate_dev(gate, ca[0], ca[1], cb[0], cb[1], want[0], want[1], third[0], third[1], want[2])
            e.gate_dev_noalloc(gate, ca[0], ca[1], cb[0], cb[1], got[0], got[1], third[0], third[1], got[2])
            torch.cuda.synchronize()
            for g, w in zip(got, want):
                assert torch.equal(g, w), (call, gate)
            if call == 0:                              # and one ciphertext against the oracle
                wa, wb, wv = orc.gate(P, gate, K, (ca[0][0].cpu().numpy(), int(ca[1][0]), 0.0),
                                      (cb[0][0].cpu().numpy(), int(cb[1][0]), 0.0), None, orc.MUL_EXACT)
                assert (got[0][0].cpu().numpy() == wa).all() and int(got[1][0]) == wb and float(got[2][0]) == wv
        # beyond the fixed array: refused, nothing launched
        B = 4096
        ca, cb = tens(3500, B), tens(3600, B)
        o = outs(B)
        with pytest.raises(tfhe_amd.TfheHipError, match="fixed scratch"):
            e.gate_dev_noalloc("nand", ca[0], ca[1], cb[0], cb[1], o[0], o[1])
    finally:
        e.close()
