#!/usr/bin/env python3
"""bench.py — bootstrapped NAND gates per second on MI355X.

A step is one batch of 16 384 independent bootstrapped NAND gates (gate
prologue + blind rotate + sample extract + key switch) on the reference's bench
parameters (Bit80: n=500, N=1024, k=1, l=2, Bg=2^10, t=8, base 4;
/root/reference/benches/bootstrapping.rs:10-25).  Inputs, outputs and keys are
resident in HBM before the timed region.  Keys are synthetic (every entry
uniform), so no stage degenerates to zeros; the kernels are data independent.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--level 80|128]
                  [--workload nand|add32|polymul]

--workload nand (default) is the headline metric; its line also carries the
blind-rotate roofline (live HIP-event durations; traffic from the committed
counter pass), p50 single-gate latency, a check of one timed output against the
CPU oracle, the CPU baseline (the restated reference algorithm on every CPU this
job may use) and, on one GPU, four more blocks, each with its own check against
the oracle and a flag that skips it:
  "bit128"   the same batch on the crate's default parameters       (--no-bit128)
  "polymul"  65 536 isolated polynomial products                    (--no-polymul)
  "add32"    4 096 level-batched 32-bit ripple-carry additions      (--no-add32)
  "textbook" the same NAND batch on libtfhe_hip_textbook.so, the textbook-correct
             variant, with keys it generates itself                 (--no-textbook)
--workload add32 / polymul time those workloads alone (--batch instances or
products per step) and print their own single JSON line.
--single-process --devices 0,1,..: one process drives the listed GPUs through
tfhe_hip_create_multi and the host-pointer entry points (what a single-threaded caller
of the crate's API would use) instead of one process per GPU.

For N > 1 launch with torch.distributed.run (one rank per GPU); rank 0 builds
the keys, their FFT-domain form is broadcast once over RCCL, and every rank
runs its own batch (weak scaling, no data-path collective).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FP64 work per blind-rotate iteration (real 2N-point FFT = 2.5*2N*log2(2N)
# flop, pointwise product on N+1 bins = 6 flop each): (k+1) l forward transforms and
# (k+1)^2 l (pointwise + inverse).
FLOP_FFT = 2.5 * 2048 * 11
FLOP_POINTWISE = 6 * 1025
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X datasheet, vector FP64
# what a stream of independent v_fma_f64 reaches on the GPU box at this kernel's occupancy (one
# wavefront per SIMD; tools/valu_cost_ubench.hip, profiles/r02_valu_cost_ubench.txt): reported
# beside the datasheet peak, never used for `frac`
FP64_MEASURED_FMA_STREAM_TFLOPS = 58.7


def flop_per_iteration(l):
    return 2 * l * FLOP_FFT + 4 * l * (FLOP_POINTWISE + FLOP_FFT)


BLIND_ROTATE_MODE = "auto"     # --mode; auto picks blind_rotate64_kernel for batches of this size

LEVELS = {80: dict(n=500, l=2, bg_bit=10), 128: dict(n=630, l=3, bg_bit=7)}


def synthetic_keys(level, seed):
    p = LEVELS[level]
    rng = np.random.default_rng(seed)
    n, l = p["n"], p["l"]

    def u(shape):
        return rng.integers(-2**31, 2**31, size=shape, dtype=np.int64).astype(np.int32)

    return dict(bk=u((n, l, 2, 2, 1024)), ks_a=u((n, 8, 4, n)), ks_b=u((n, 8, 4)),
                ks_var=rng.random((n, 8, 4)) * 1e-9)


def reference_style_keys(level, seed):
    """Keys with the structure the crate's generate_keys produces (src/lwe.rs:474-558,
    src/tgsw.rs:118-167): in every TGSW sample only the rows of decomposition level 0 are
    encrypted (uniform mask, one Gaussian coefficient in the body's first N entries) and
    the secret bit times l is added to coefficient 0 of each polynomial; key-switch
    entries of digit 0 are zero.  Used to show that throughput does not depend on the key
    material (the kernels are data independent)."""
    p = LEVELS[level]
    rng = np.random.default_rng(seed)
    n, l = p["n"], p["l"]
    key = rng.integers(0, 2, size=n, dtype=np.int64)
    bk = np.zeros((n, l, 2, 2, 1024), dtype=np.int64)
    bk[:, 0, :, 0, :] = rng.integers(-2**31, 2**31, size=(n, 2, 1024))
    bk[:, 0, :, 1, 1023] = np.rint(rng.normal(0.0, 9.0e-9 * 2.0**32, size=(n, 2)))
    bk[:, 0, :, :, 0] += (key * l)[:, None, None]
    ks_a = rng.integers(-2**31, 2**31, size=(n, 8, 4, n), dtype=np.int64)
    ks_b = rng.integers(-2**31, 2**31, size=(n, 8, 4), dtype=np.int64)
    ks_var = np.full((n, 8, 4), 1e-10)
    ks_a[:, :, 0], ks_b[:, :, 0], ks_var[:, :, 0] = 0, 0, 0.0
    wrap = lambda x: ((x + 2**31) % 2**32 - 2**31).astype(np.int32)
    return dict(bk=wrap(bk), ks_a=wrap(ks_a), ks_b=wrap(ks_b), ks_var=ks_var)


def measured_traffic(level, batch):
    """HBM-side bytes per blind-rotate launch from the committed PMC passes of this very
    command line (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, corrected as the
    CDNA4 guide prescribes; tools/collect_profiles.py writes the file).  Counters cannot be
    read inside the timed run, so this is a profiled value: only valid for the workload it was
    collected on, otherwise null."""
    for name in ("r03_traffic.json",):
        path = os.path.join(ROOT, "profiles", name)
        if level != 80 or batch != 16384 or not os.path.exists(path):
            continue
        try:
            return float(json.load(open(path))["traffic_bytes_per_launch"]), "profiles/" + name
        except Exception:  # noqa: BLE001
            continue
    return None, None


def host_threads():
    """Threads for the CPU baseline: the CPUs this process may run on, capped by the share
    the GPU box grants a one-GPU job (it exports that share as OMP_NUM_THREADS; nproc there
    shows the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(level):
    """Times the CPU restatement of tfhe_bootstrap (the checker, oracle/) on the host."""
    exe = os.path.join(ROOT, "oracle", "cpu_baseline")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("cpu_baseline.cpp", "oracle.cpp", "oracle.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(x) for x in srcs):
        try:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_baseline"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:  # noqa: BLE001
            pass
    if not os.path.exists(exe):
        return None
    cores = host_threads()
    per_thread = 2
    try:
        out = subprocess.run([exe, str(level), str(cores), str(per_thread)], capture_output=True,
                             text=True, timeout=600)
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"value": None, "error": repr(e)}
    return {"value": r["gates_per_sec"], "unit": "gates/s", "cores": cores, "kind": "port",
            "per_core": r["gates_per_sec"] / cores,
            "sample": f"{r['gates']} tfhe_bootstrap calls ({per_thread} per thread, {cores} threads "
                      f"at once = every CPU this job may use; os.cpu_count() = {os.cpu_count()}) on "
                      f"reference-style Bit{level} keys; C++ restatement of the "
                      f"reference algorithm with a fresh FFT plan per product as in "
                      f"src/numerics.rs:235-251, not the Rust binary"}


def oracle_check(level, K, ca_a, ca_b, cb_a, cb_b, out_a, out_b, index):
    """One gate of the timed batch against the CPU oracle (the checker; never timed)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    P = orc.params(level)
    Ko = {k: np.ascontiguousarray(v) for k, v in K.items()}
    a = (ca_a[index].cpu().numpy(), int(ca_b[index]), 0.0)
    b = (cb_a[index].cpu().numpy(), int(cb_b[index]), 0.0)
    wa, wb, _ = orc.gate(P, "nand", Ko, a, b, None, orc.MUL_FFT)
    return bool((out_a[index].cpu().numpy() == wa).all() and int(out_b[index]) == wb)


def progress(msg):
    """Progress lines go to stderr; stdout carries the one JSON line."""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def init_ranks(torch, dist):
    """One process per GPU: reads RANK / LOCAL_RANK / WORLD_SIZE, selects the GPU, joins the
    process group (RCCL).  Returns (world, rank, device).
    Rehearsal of the multi-rank path on a box with fewer GPUs than ranks: TFHE_BENCH_REHEARSE=1
    puts every rank on GPU 0 and uses gloo (RCCL refuses two ranks on one GPU); the numbers of
    such a run mean nothing, it only exercises the code path."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("TFHE_BENCH_REHEARSE") == "1"
    gpu = 0 if (rehearse or world == 1) else local_rank
    torch.cuda.set_device(gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
    return world, rank, torch.device("cuda", gpu)


def side_workload(args):
    """Secondary workloads (not the headline metric): one JSON line each."""
    import torch
    import torch.distributed as dist

    import tfhe_amd

    world, rank, dev = init_ranks(torch, dist)
    lv = LEVELS[args.level]
    eng = tfhe_amd.Engine(dev.index)
    eng.use_torch_stream()
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + rank)

    def rnd(shape, lo=-2**31, hi=2**31):
        return torch.randint(lo, hi, shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.workload == "add32":
        params = tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"])
        if rank == 0:
            K = synthetic_keys(args.level, 0xC0FFEE + 5)
            eng.load_keys(params, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        if world > 1:
            from tfhe_amd import parallel
            parallel.replicate_keys(eng, params, dev, src=0)
        count = args.batch if args.batch != 16384 else 4096
        bits, n = 32, lv["n"]
        a_a, a_b, b_a, b_b = rnd((bits, count, n)), rnd((bits, count)), rnd((bits, count, n)), rnd((bits, count))
        s_a = torch.empty((bits, count, n), dtype=torch.int32, device=dev)
        s_b = torch.empty((bits, count), dtype=torch.int32, device=dev)
        c_a = torch.empty((count, n), dtype=torch.int32, device=dev)
        c_b = torch.empty((count,), dtype=torch.int32, device=dev)

        def step():
            eng.circuit_add_dev(a_a, a_b, b_a, b_b, s_a, s_b, c_a, c_b)
        units, unit_name = 5 * bits * count, "gates/s"
        workload = (f"circuits::add, {count} instances of {bits}-bit operands per GPU per step "
                    f"(160 bootstrapped gates each, 65 dependency levels), Bit{args.level}")
        metric = "bootstrapped gates/sec inside level-batched 32-bit ripple-carry adders"
    else:
        B = args.batch if args.batch != 16384 else 65536
        half = 512 if args.level == 80 else 64
        d, s_ = rnd((B, 1024), -half, half), rnd((B, 1024))
        out = torch.empty((B, 2047), dtype=torch.int32, device=dev)

        def step():
            eng.polymul_dev(d, s_, out)
        units, unit_name = B, "products/s"
        workload = (f"{B} independent products of a digit polynomial (|d| < {half}) with a uniform "
                    f"Torus32 polynomial, N = 1024, full 2047-coefficient result")
        metric = "isolated polynomial products/sec (poly_multiplier)"

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        value = units * world * args.steps / elapsed
        res = {"metric": metric, "value": value, "unit": unit_name, "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": workload}}
        if args.workload == "polymul":
            flop = 2 * FLOP_FFT + FLOP_FFT + FLOP_POINTWISE
            bytes_ = 1024 * 4 * 2 + 2047 * 4
            res["roofline"] = {"bound": "fp64_valu", "achieved": value * flop / 1e12,
                               "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": value * flop / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                               "hbm_GBps_algorithmic": value * bytes_ / 1e9}
        if args.workload == "add32":
            res["additions_per_sec"] = value / (5 * 32)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def polymul_block(eng, torch, dev, level, B=65536, steps=300, warmup=100):
    """BASELINE config 2: B isolated products; a sample of them against the exact oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    half = 512 if level == 80 else 64
    d = torch.randint(-half, half, (B, 1024), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    s_ = torch.randint(-2**31, 2**31, (B, 1024), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    out = torch.empty((B, 2047), dtype=torch.int32, device=dev)
    for _ in range(warmup):
        eng.polymul_dev(d, s_, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.polymul_dev(d, s_, out)
    torch.cuda.synchronize()
    value = B * steps / (time.perf_counter() - t0)
    ok = True
    for i in (0, B // 3, B - 1):
        want, _ = orc.poly_multiplier(d[i].cpu().numpy(), s_[i].cpu().numpy(), orc.MUL_EXACT)
        ok = ok and bool((out[i].cpu().numpy() == want).all())
    flop = 3 * FLOP_FFT + FLOP_POINTWISE
    return {"metric": "isolated polynomial products/sec (poly_multiplier)", "value": value, "unit": "products/s",
            "config": f"{B} independent products of a digit polynomial (|d| < {half}) with a uniform Torus32 "
                      f"polynomial, N = 1024, full 2047-coefficient result",
            "roofline": {"bound": "fp64_valu", "kernel": "polymul64_kernel", "achieved": value * flop / 1e12,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": value * flop / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "hbm_GBps_algorithmic": value * (2 * 4096 + 2047 * 4) / 1e9},
            "output_check": {"products": [0, B // 3, B - 1], "matches_oracle": ok}}


def add32_block(eng, torch, dev, level, K, count=4096, bits=32):
    """BASELINE config 5: `count` instances of circuits::add on `bits`-bit operands, one step;
    the two lowest sum bits of one instance against the same gates composed on the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    n = LEVELS[level]["n"]
    g = torch.Generator(device=dev)
    g.manual_seed(4322)

    def rnd(shape):
        return torch.randint(-2**31, 2**31, shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32)

    a_a, a_b, b_a, b_b = rnd((bits, count, n)), rnd((bits, count)), rnd((bits, count, n)), rnd((bits, count))
    s_a = torch.empty((bits, count, n), dtype=torch.int32, device=dev)
    s_b = torch.empty((bits, count), dtype=torch.int32, device=dev)
    c_a = torch.empty((count, n), dtype=torch.int32, device=dev)
    c_b = torch.empty((count,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.circuit_add_dev(a_a, a_b, b_a, b_b, s_a, s_b, c_a, c_b)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # full_adder of bits 0 and 1 of one instance (src/circuits/circuits.rs:83-97) on the oracle
    P = orc.params(level)
    Ko = {k: np.ascontiguousarray(v) for k, v in K.items()}
    c = count // 2
    smp = lambda ta, tb, i: (ta[i, c].cpu().numpy(), int(tb[i, c]), 0.0)
    gate = lambda name, x, y: orc.gate(P, name, Ko, x, y, None, orc.MUL_FFT)
    carry = (np.zeros(n, np.int32), -(1 << 29), 0.0)                       # constant(false)
    ok = True
    for i in range(2):
        x, y = smp(a_a, a_b, i), smp(b_a, b_b, i)
        axb = gate("xor", x, y)
        want = gate("xor", axb, carry)
        ok = ok and bool((s_a[i, c].cpu().numpy() == want[0]).all() and int(s_b[i, c]) == want[1])
        if i == 0:
            carry = gate("or", gate("and", carry, axb), gate("and", x, y))
    gates = 5 * bits * count
    return {"metric": "bootstrapped gates/sec inside level-batched ripple-carry adders", "value": gates / elapsed,
            "unit": "gates/s", "additions_per_sec": count / elapsed, "seconds": elapsed,
            "config": f"circuits::add, {count} instances of {bits}-bit operands in one call ({5 * bits} bootstrapped "
                      f"gates each, {2 * bits + 1} dependency levels), Bit{level}",
            "output_check": {"instance": c, "sum_bits": [0, 1], "matches_oracle": ok}}


def textbook_block(torch, dev, level, B, steps):
    """The same NAND batch on libtfhe_hip_textbook.so (the textbook-correct variant: negacyclic
    products summed in the frequency domain, all TGSW rows encrypted, N-row key switch), with keys
    from its own tfhe_hip_generate_keys and fresh encryptions as inputs: gates/s, the blind-rotate
    roofline on the same algorithmic convention ((k+1) l forward, (k+1) inverse transforms and
    (k+1)^2 l pointwise products per iteration), every output of a sample decrypted and one
    gate compared with the exact oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    import tfhe_amd
    eng = tfhe_amd.Engine(dev.index, variant="textbook")
    try:
        eng.use_torch_stream()
        lv = LEVELS[level]
        P = orc.tb_params(level, bk_stdev=2.0 ** -25 if level == 128 else None)
        hp = tfhe_amd.make_params(P.n, P.N, P.k, P.l, P.bg_bit, P.ks_t, P.ks_base_bit, P.n_ks)
        K = eng.generate_keys(hp, P.ks_stdev, P.bk_stdev, bytes(range(32)))
        # 64 fresh encryptions, tiled over the batch
        bits = [((i >> 1) & 1, i & 1) for i in range(64)]
        ca = [orc.encrypt(P, K["lwe_key"], x, 100 + i) for i, (x, _) in enumerate(bits)]
        cb = [orc.encrypt(P, K["lwe_key"], y, 200 + i) for i, (_, y) in enumerate(bits)]
        rep = (B + 63) // 64
        tile = lambda cs, k: torch.from_numpy(np.stack([c[k] for c in cs])).to(dev).repeat((rep,) + (1,) * (k == 0))[:B].contiguous()
        ca_a, cb_a = tile(ca, 0), tile(cb, 0)
        ca_b = torch.tensor([c[1] for c in ca], dtype=torch.int32, device=dev).repeat(rep)[:B].contiguous()
        cb_b = torch.tensor([c[1] for c in cb], dtype=torch.int32, device=dev).repeat(rep)[:B].contiguous()
        out_a = torch.empty((B, P.n), dtype=torch.int32, device=dev)
        out_b = torch.empty((B,), dtype=torch.int32, device=dev)
        eng.gate_dev("nand", ca_a, ca_b, cb_a, cb_b, out_a, out_b)
        torch.cuda.synchronize()
        eng.profile_enable(True)
        eng.profile_read()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.gate_dev("nand", ca_a, ca_b, cb_a, cb_b, out_a, out_b)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        prof = eng.profile_read()
        br_ms = prof["blind_rotate"][0] / max(prof["blind_rotate"][1], 1)
        flop = (2 * lv["l"] * FLOP_FFT + 2 * FLOP_FFT + 4 * lv["l"] * FLOP_POINTWISE) * lv["n"] * B
        oa, ob = out_a[:64].cpu().numpy(), out_b[:64].cpu().numpy()
        decrypted = all(orc.decrypt(oa[i], ob[i], K["lwe_key"]) == (not (x and y)) for i, (x, y) in enumerate(bits))
        Ko = {k: np.ascontiguousarray(v) for k, v in K.items()}
        wa, wb, _ = orc.tb_gate(P, "nand", Ko, ca[3][:2], cb[3][:2])
        return {"metric": "bootstrapped NAND gates/sec, textbook-correct variant (libtfhe_hip_textbook.so)",
                "value": B * steps / elapsed, "unit": "gates/s",
                "config": f"{B} NAND gates per step, Bit{level} shape, keys from tfhe_hip_generate_keys, inputs "
                          f"fresh encryptions; not the reference's algorithm (see tfhe_hip_variant in the header)",
                "kernels_ms_per_launch": {"blind_rotate": br_ms,
                                          "keyswitch": prof["keyswitch"][0] / max(prof["keyswitch"][1], 1)},
                "roofline": {"bound": "fp64_valu", "kernel": "blind_rotate_textbook_kernel",
                             "achieved": flop / (br_ms / 1e3) / 1e12 if br_ms else 0.0,
                             "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": flop / (br_ms / 1e3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS if br_ms else 0.0,
                             "flop_per_launch": flop},
                "output_check": {"decrypts_to_nand_truth_table": bool(decrypted), "samples": 64,
                                 "matches_oracle": bool((oa[3] == wa).all() and int(ob[3]) == wb)}}
    finally:
        eng.close()


def time_nand(eng, torch, dist, dev, world, rank, level, K, B, steps, warmup, check=True, chunk=None,
              total=None):
    """Times `steps` batches of B NAND gates on the loaded keys (each batch in launches of at
    most `chunk` gates); returns the measurements of this rank (max over ranks for the elapsed
    time) and checks three timed outputs (first, a third in, last) against the oracle.  total: gates of all ranks per step (default B * world)."""
    chunk = chunk or B
    lv = LEVELS[level]
    n = lv["n"]
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank + level)

    def rnd(shape):
        return torch.randint(-2**31, 2**31, shape, dtype=torch.int64, device=dev,
                             generator=g).to(torch.int32)

    ca_a, ca_b, cb_a, cb_b = rnd((B, n)), rnd((B,)), rnd((B, n)), rnd((B,))
    out_a = torch.empty((B, n), dtype=torch.int32, device=dev)
    out_b = torch.empty((B,), dtype=torch.int32, device=dev)

    def step():
        for c in range(0, B, chunk):
            e = min(B, c + chunk)
            eng.gate_dev("nand", ca_a[c:e], ca_b[c:e], cb_a[c:e], cb_b[c:e], out_a[c:e], out_b[c:e])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    eng.profile_enable(True)
    eng.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    per_rank = [B * steps / elapsed]
    if world > 1:
        cdev = "cpu" if dist.get_backend() == "gloo" else dev
        mine = torch.tensor([B * steps / elapsed], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(x.item()) for x in every]
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    br_ms, br_cnt = prof["blind_rotate"]
    ks_ms, ks_cnt = prof["keyswitch"]
    br_avg_s = br_ms / max(br_cnt, 1) / 1e3
    # algorithmic flop of the timed launches over their summed duration
    flop_launch = flop_per_iteration(lv["l"]) * lv["n"] * B * steps / max(br_cnt, 1)
    achieved = flop_launch / br_avg_s / 1e12 if br_avg_s > 0 else 0.0
    traffic, traffic_from = measured_traffic(level, chunk)
    res = {
        "value": (total if total is not None else B * world) * steps / elapsed,
        "ms_per_step": elapsed / steps * 1e3,
        "per_rank_value": per_rank,
        "roofline": {"bound": "fp64_valu",
                     # the throughput form the library's AUTO mode picks for this parameter set
                     "kernel": {"batch": "blind_rotate_kernel", "split": "blind_rotate_split64_kernel"}.get(
                         BLIND_ROTATE_MODE, "blind_rotate64_kernel"),
                     "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                     "measured_fma_stream_peak": FP64_MEASURED_FMA_STREAM_TFLOPS,
                     "traffic": traffic, "traffic_profiled_from": traffic_from,
                     "flop_per_launch": flop_launch, "avg_launch_ms": br_avg_s * 1e3,
                     "launches": br_cnt},
        "kernels_ms_per_launch": {"blind_rotate": br_ms / max(br_cnt, 1),
                                  "keyswitch": ks_ms / max(ks_cnt, 1),
                                  "prologue": prof["prologue"][0] / max(prof["prologue"][1], 1)},
    }
    if check and rank == 0 and K is not None:
        # the outputs of the last timed step: one gate against the CPU oracle
        idx = sorted({0, B // 3, B - 1})
        res["output_check"] = {"gate_indices": idx, "matches_oracle": all(
            oracle_check(level, K, ca_a, ca_b, cb_a, cb_b, out_a, out_b, i) for i in idx)}
    return res, (ca_a, ca_b, cb_a, cb_b, out_a, out_b)


def latency(eng, torch, tensors, K_params_n):
    """p50 of one NAND (B = 1), host-observed submit-to-completion: through the device-pointer
    entry point (operands resident in HBM) and through the host-pointer entry point (what a
    single tfhe_bootstrap-style call from the reference's API pays: staging both ways)."""
    ca_a, ca_b, cb_a, cb_b, out_a, out_b = tensors
    lat = []
    for _ in range(201):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.gate_dev("nand", ca_a[:1], ca_b[:1], cb_a[:1], cb_b[:1], out_a[:1], out_b[:1])
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    p50_dev = sorted(lat[1:])[len(lat[1:]) // 2] * 1e3
    ha = (ca_a[:1].cpu().numpy(), ca_b[:1].cpu().numpy())
    hb = (cb_a[:1].cpu().numpy(), cb_b[:1].cpu().numpy())
    lat = []
    for _ in range(101):
        t1 = time.perf_counter()
        eng.gate("nand", ha, hb)
        lat.append(time.perf_counter() - t1)
    p50_host = sorted(lat[1:])[len(lat[1:]) // 2] * 1e3
    return p50_dev, p50_host


def single_process(args):
    """One process, many GPUs: tfhe_hip_create_multi over --devices, keys loaded once and
    replicated by the library (RCCL broadcast or peer copies), every step one host-pointer
    tfhe_hip_gate call of B gates PER DEVICE that the library splits contiguously and runs at
    the same time.  Reports the aggregate rate (host staging included: this path hands over host
    buffers) and, measured in the same process, the rate of the first device alone."""
    import tfhe_amd
    devices = [int(x) for x in (args.devices or ",".join(str(i) for i in range(args.gpus))).split(",")]
    lv = LEVELS[args.level]
    params = tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"])
    K = synthetic_keys(args.level, 0xC0FFEE + 3)
    rng = np.random.default_rng(99)

    def run(devs):
        eng = tfhe_amd.Engine(devs if len(devs) > 1 else devs[0])
        try:
            eng.load_keys(params, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
            B = args.batch * len(devs)
            u = lambda shape: rng.integers(-2**31, 2**31, size=shape, dtype=np.int64).astype(np.int32)
            ca, cb = (u((B, lv["n"])), u((B,))), (u((B, lv["n"])), u((B,)))
            for _ in range(args.warmup):
                out = eng.gate("nand", ca, cb)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = eng.gate("nand", ca, cb)
            elapsed = time.perf_counter() - t0
            transport = eng.key_transport() if len(devs) > 1 else "none"
            return B * args.steps / elapsed, elapsed / args.steps * 1e3, transport, (ca, cb, out)
        finally:
            eng.close()

    value, ms, transport, (ca, cb, out) = run(devices)
    single = run(devices[:1])[0] if len(devices) > 1 else value
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    P = orc.params(args.level)
    i = out[1].size - 1                                              # a gate of the last device's shard
    wa, wb, _ = orc.gate(P, "nand", {k: np.ascontiguousarray(v) for k, v in K.items()},
                         (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0), None, orc.MUL_FFT)
    res = {"metric": "bootstrapped NAND gates/sec (node, default params)", "value": value, "unit": "gates/s",
           "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"one process, tfhe_hip_create_multi over devices {devices}: {args.batch} NAND gates "
                                  f"per device per step through the host-pointer entry point (PCIe staging "
                                  f"included), Bit{args.level}, synthetic keys replicated by the library",
                      "batch_per_gpu": args.batch, "security_level": args.level, "key_transport": transport,
                      "parallelism": "one host thread per device inside the library, batch split contiguously"},
           "single_device_value": single,
           "output_check": {"gate_index": int(i), "matches_oracle": bool((out[0][i] == wa).all() and int(out[1][i]) == wb)},
           "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(args.level)}
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 3 (polymul: 1000)")
    ap.add_argument("--warmup", type=int, default=None, help="default 1 (polymul: 200)")
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--level", type=int, default=80, choices=[80, 128])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-bit128", action="store_true",
                    help="skip the Bit128 block that the default Bit80 run adds")
    ap.add_argument("--no-polymul", action="store_true", help="skip the isolated-product block")
    ap.add_argument("--no-add32", action="store_true", help="skip the 32-bit adder block")
    ap.add_argument("--no-textbook", action="store_true", help="skip the textbook-variant block")
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives every GPU of --devices through tfhe_hip_create_multi")
    ap.add_argument("--devices", default=None, help="comma-separated GPU ordinals for --single-process")
    ap.add_argument("--total-gates", type=int, default=None,
                    help="strong scaling: this many gates per step over ALL ranks (contiguous shards, "
                         "launches of --batch gates); BASELINE config 4 is --total-gates 1048576")
    ap.add_argument("--workload", default="nand", choices=["nand", "add32", "polymul"])
    ap.add_argument("--mode", default="auto", choices=["auto", "batch", "split", "batch64"],
                    help="GPU mapping of the blind rotation (tfhe_hip_set_blind_rotate_mode); results are "
                         "bit-identical in every mode, auto is what a caller gets")
    ap.add_argument("--keys", default="synthetic", choices=["synthetic", "reference"],
                    help="synthetic: every key entry uniform (no stage degenerates); reference: "
                         "the structure the crate's generate_keys produces")
    args = ap.parse_args()
    # a polymul step is ~0.6 ms and the first ~100 ms of a burst run at lower clocks (-6 %)
    dsteps, dwarm = (1000, 200) if args.workload == "polymul" else (3, 1)
    args.steps = dsteps if args.steps is None else args.steps
    args.warmup = dwarm if args.warmup is None else args.warmup
    if args.workload != "nand":
        return side_workload(args)
    if args.single_process:
        return single_process(args)

    import torch
    import torch.distributed as dist

    import tfhe_amd

    world, rank, dev = init_ranks(torch, dist)

    eng = tfhe_amd.Engine(dev.index)
    eng.use_torch_stream()
    if args.mode != "auto":
        global BLIND_ROTATE_MODE
        BLIND_ROTATE_MODE = args.mode
        eng.set_blind_rotate_mode(args.mode)

    def load(level):
        """rank 0 loads and transforms, the device blobs are broadcast once"""
        lv = LEVELS[level]
        params = tfhe_amd.make_params(lv["n"], l=lv["l"], bg_bit=lv["bg_bit"])
        K = None
        if rank == 0:
            K = (synthetic_keys if args.keys == "synthetic" else reference_style_keys)(level, 0xC0FFEE + 3)
            eng.load_keys(params, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        progress(f"Bit{level} keys loaded on rank 0")
        if world > 1:
            from tfhe_amd import parallel
            parallel.replicate_keys(eng, params, dev, src=0)
            progress(f"keys replicated to {world} ranks")
        return K

    B = args.batch
    lv = LEVELS[args.level]
    K = load(args.level)
    if args.total_gates:
        from tfhe_amd import parallel
        lo, hi = parallel.shard_bounds(args.total_gates, rank, world)
        main_res, tensors = time_nand(eng, torch, dist, dev, world, rank, args.level, K, hi - lo, args.steps,
                                      args.warmup, chunk=B, total=args.total_gates)
    else:
        main_res, tensors = time_nand(eng, torch, dist, dev, world, rank, args.level, K, B, args.steps,
                                      args.warmup)
    progress(f"timed {args.steps} steps: {main_res['value']:.0f} gates/s")
    p50 = p50_host = None
    if rank == 0 and not args.no_latency:
        p50, p50_host = latency(eng, torch, tensors, lv["n"])
    del tensors

    # Bit128 (the crate's Parameters::default, src/lwe.rs:235-241) beside the bench's Bit80
    bit128 = None
    if args.level == 80 and not args.no_bit128:
        K128 = load(128)
        bit128, t128 = time_nand(eng, torch, dist, dev, world, rank, 128, K128, B, max(2, min(args.steps, 3)), 1)
        del t128, K128
        if rank == 0:
            bit128["unit"] = "gates/s"
            bit128["config"] = "same batch on Bit128 (n=630, N=1024, k=1, l=3, Bg=2^7, ks t=8 base=4)"

    # BASELINE configs 2 and 5 and the textbook-correct variant beside the headline (one GPU only)
    blocks = {}
    if world == 1 and not args.total_gates:
        if not args.no_polymul:
            blocks["polymul"] = polymul_block(eng, torch, dev, args.level)
            progress(f"polymul block: {blocks['polymul']['value'] / 1e6:.1f} M products/s")
        if not args.no_add32:
            if bit128 is not None:
                K = load(args.level)                       # the Bit128 block replaced the keys
            blocks["add32"] = add32_block(eng, torch, dev, args.level, K)
            progress(f"add32 block: {blocks['add32']['value']:.0f} gates/s")
        if not args.no_textbook:
            blocks["textbook"] = textbook_block(torch, dev, args.level, B, 2)
            progress(f"textbook block: {blocks['textbook']['value']:.0f} gates/s")

    if rank == 0:
        res = {
            "metric": "bootstrapped NAND gates/sec (node, default params)",
            "value": main_res["value"], "unit": "gates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if args.total_gates else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{args.total_gates} bootstrapped NAND gates per step in contiguous shards "
                                    f"over {world} GPU(s), launches of {B} gates, " if args.total_gates else
                                    f"batched bootstrapped NAND, {B} gates per GPU per step, ") +
                                   f"Bit{args.level} (n={lv['n']}, N=1024, k=1, l={lv['l']}, "
                                   f"Bg=2^{lv['bg_bit']}, ks t=8 base=4), {args.keys} keys resident in HBM",
                       "total_gates": args.total_gates,
                       "batch_per_gpu": B, "security_level": args.level,
                       "parallelism": f"batch sharded over {world} GPU(s), keys replicated"},
            "p50_single_gate_latency_ms": p50,
            "p50_single_gate_latency_host_pointers_ms": p50_host,
            "roofline": main_res["roofline"],
            "kernels_ms_per_launch": main_res["kernels_ms_per_launch"],
            "output_check": main_res.get("output_check"),
            "bit128": bit128,
        }
        res.update(blocks)
        if world > 1:
            res["per_rank_value"] = main_res["per_rank_value"]
        # rank 0's host cores, also for N > 1 (the other ranks wait at the barrier below)
        res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.level)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
