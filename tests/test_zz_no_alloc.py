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


@pytest.mark.parametrize("gate,B", [("nand", 600), ("mux", 40)])
def test_gate_dev_noalloc_under_stream_capture(gate, B):
    """The claim itself: nothing is allocated.  The call is recorded into a HIP graph (stream
    capture in global mode, where hipMalloc, hipFree and synchronisation are errors that
    invalidate the capture) on the context's stream, and the graph is replayed twice on fresh
    inputs: outputs equal tfhe_hip_gate_dev's.  B = 600 takes the throughput mapping (its
    per-wavefront workspace was allocated with the keys), B = 40 the one-workgroup-per-ciphertext
    mapping."""
    import torch
    import tfhe_amd
    dev = torch.device("cuda", 0)
    P = orc.params(80)
    P.n = P.n_ks = 12
    K = orc.generate_synthetic_keys(P, 3700)
    e = tfhe_amd.Engine(0)
    try:
        e.load_keys(tfhe_amd.make_params(P.n, n_ks=P.n_ks), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        stream = torch.cuda.Stream(device=dev)

        def fresh(seed):
            return [torch.from_numpy(orc.uniform_i32(seed + k, (B, P.n) if k % 2 == 0 else (B,))).to(dev)
                    for k in range(6)]

        ins = fresh(3800)
        got = (torch.empty((B, P.n), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
               torch.empty((B,), dtype=torch.float64, device=dev))
        third = (ins[4], ins[5]) if gate == "mux" else (None, None)
        with torch.cuda.stream(stream):
            e.use_torch_stream()                       # the context now enqueues on `stream`
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="global"):
            e.gate_dev_noalloc(gate, ins[0], ins[1], ins[2], ins[3], got[0], got[1], third[0], third[1], got[2])
        for round_ in range(2):
            new = fresh(3900 + 100 * round_)
            for dst, src in zip(ins, new):
                dst.copy_(src)
            for o in got:
                o.fill_(-7)
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            want = (torch.empty_like(got[0]), torch.empty_like(got[1]), torch.empty_like(got[2]))
            with torch.cuda.stream(stream):
                e.gate_dev(gate, ins[0], ins[1], ins[2], ins[3], want[0], want[1], third[0], third[1], want[2])
            torch.cuda.synchronize()
            for g, w in zip(got, want):
                assert torch.equal(g, w), (gate, round_)
    finally:
        torch.cuda.synchronize()
        e.close()


def test_the_allocating_entry_point_is_refused_under_the_same_capture():
    """Control for the test above, in a process of its own (a failed capture leaves a sticky
    error behind): tfhe_hip_gate_dev on a fresh context, whose scratch pool is still empty and
    must be allocated, cannot be recorded."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import torch, _oracle as orc, tfhe_amd\n"
        "dev = torch.device('cuda', 0)\n"
        "P = orc.params(80); P.n = P.n_ks = 12; K = orc.generate_synthetic_keys(P, 3700)\n"
        "e = tfhe_amd.Engine(0)\n"
        "e.load_keys(tfhe_amd.make_params(P.n, n_ks=P.n_ks), K['bk'], K['ks_a'], K['ks_b'], K['ks_var'])\n"
        "B = 600\n"
        "t = [torch.from_numpy(orc.uniform_i32(k, (B, P.n) if k % 2 == 0 else (B,))).to(dev) for k in range(4)]\n"
        "oa = torch.empty((B, P.n), dtype=torch.int32, device=dev); ob = torch.empty((B,), dtype=torch.int32, device=dev)\n"
        "s = torch.cuda.Stream(device=dev)\n"
        "with torch.cuda.stream(s):\n"
        "    e.use_torch_stream()\n"
        "torch.cuda.synchronize()\n"
        "g = torch.cuda.CUDAGraph()\n"
        "try:\n"
        "    with torch.cuda.graph(g, stream=s, capture_error_mode='global'):\n"
        "        e.gate_dev('nand', t[0], t[1], t[2], t[3], oa, ob)\n"
        "    print('RECORDED')\n"
        "except Exception as x:\n"
        "    print('REFUSED', type(x).__name__)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert "REFUSED" in r.stdout and "RECORDED" not in r.stdout, r.stdout + r.stderr[-2000:]
