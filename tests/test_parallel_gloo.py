"""The N > 1 path on CPU: two processes over gloo exercise the key-blob
broadcast and the static gate sharding that bench.py uses with RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_shard_bounds_cover_everything():
    from tfhe_amd.parallel import shard_bounds
    for total in (0, 1, 7, 16384, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


class _FakeEngine:
    """Stands in for tfhe_amd.Engine: same blob export/import calls, CPU tensors."""

    def __init__(self, rank):
        self.rank = rank
        self.imported = None

    def key_blob_sizes(self, params):
        return 4096, 1000

    def export_keys_dev(self, bk_blob, ks_blob):
        bk_blob.copy_(torch.arange(bk_blob.numel(), dtype=torch.int64).remainder(251).to(torch.uint8))
        ks_blob.copy_(torch.arange(ks_blob.numel(), dtype=torch.int64).remainder(13).to(torch.uint8))

    def import_keys_dev(self, params, bk_blob, ks_blob):
        self.imported = (bk_blob.clone(), ks_blob.clone())


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfhe_amd import parallel
        eng = _FakeEngine(rank)
        nbytes = parallel.replicate_keys(eng, params=None, device=torch.device("cpu"), src=0)
        assert nbytes == 5096
        want_bk = torch.arange(4096, dtype=torch.int64).remainder(251).to(torch.uint8)
        want_ks = torch.arange(1000, dtype=torch.int64).remainder(13).to(torch.uint8)
        if rank == 0:
            assert eng.imported is None            # the source keeps its own keys
        else:
            assert torch.equal(eng.imported[0], want_bk) and torch.equal(eng.imported[1], want_ks)
        # sharded "gates": every rank processes its slice, the union is the whole batch
        total = 1001
        lo, hi = parallel.shard_bounds(total, rank, world)
        mine = torch.zeros(total, dtype=torch.int64)
        mine[lo:hi] = 1
        dist.all_reduce(mine)
        assert bool((mine == 1).all())
        # max-over-ranks timing as in bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_key_broadcast_and_sharding():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
