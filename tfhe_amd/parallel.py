"""Multi-GPU layer: one process per GPU, gates sharded, keys broadcast once.

The bootstrapped-gate path has no exchange step: every gate of a batch is
independent.  The only collective is the one-time broadcast of the two device
key blobs (FFT-domain bootstrapping key, packed key-switch key) from the rank
that built them; it runs on whatever backend the process group uses ("nccl",
i.e. RCCL over xGMI, on GPUs; "gloo" on CPU for tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous static shard [lo, hi) of `total` gates for `rank` of `world`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blobs(blobs, src=0, group=None):
    """Broadcasts a list of uint8 tensors in place from rank `src`."""
    staged = dist.get_backend(group) == "gloo"      # gloo moves host memory: stage device blobs
    for b in blobs:
        if b.dtype != torch.uint8 or not b.is_contiguous():
            raise ValueError("blobs must be contiguous uint8 tensors")
        if staged and b.device.type != "cpu":
            h = b.cpu()
            dist.broadcast(h, src, group=group)
            b.copy_(h)
        else:
            dist.broadcast(b, src, group=group)
    return blobs


def replicate_keys(engine, params, device, src=0, group=None):
    """After `engine.load_keys` on rank `src`, gives every rank the same device keys.

    engine: tfhe_amd.Engine of this rank; params: tfhe_hip_params (same on all ranks).
    src is a GLOBAL rank (as torch.distributed.broadcast takes it), also when `group` is a
    sub-group.
    """
    rank = dist.get_rank()          # global, to compare with src
    if group is not None and src not in dist.get_process_group_ranks(group):
        raise ValueError("src must be a member of the group")
    bk_bytes, ks_bytes = engine.key_blob_sizes(params)
    bk_blob = torch.empty(bk_bytes, dtype=torch.uint8, device=device)
    ks_blob = torch.empty(ks_bytes, dtype=torch.uint8, device=device)
    if rank == src:
        engine.export_keys_dev(bk_blob, ks_blob)
    broadcast_blobs([bk_blob, ks_blob], src, group)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if rank != src:
        engine.import_keys_dev(params, bk_blob, ks_blob)
    return bk_bytes + ks_bytes
