"""Bin-sharded packing of ONE archive over several ranks (one process per GPU).

Bins are independent: every rank codes its share of the standard bins -- longest-processing-time-first over
the per-signature record totals of the .bmeta footer, the same table on every rank -- and holds the blocks (rank 0 also
the merged small-bins/N block).  The only exchange is ONE all-reduce of the block-size table (a u64 per block of the
archive: RCCL on GPUs, gloo in the CPU tests): each rank contributes its own sizes, zeros elsewhere.  From the summed table
every rank derives the offsets of its blocks in the final `.cdata` (block 0, then ascending signature: the reference's
-t1 order) and writes them there itself; rank 0 writes the `.cmeta`.  No block bytes cross ranks.
"""
import os

import numpy as np


def pack_sharded(packer, in_prefix, out_prefix, dist, device=None):
    """packer: fastore_amd.Packer created with rank=dist.get_rank(), world_size=dist.get_world_size().
    device: torch device of the collective's tensor (the rank's GPU under nccl = RCCL; None = CPU for gloo)."""
    import torch
    _, sizes = packer.shard_pack(in_prefix)
    t = torch.from_numpy(sizes.astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                    # the one collective of the path
    all_sizes = t.cpu().numpy().astype(np.uint64)
    packer.shard_write(out_prefix, all_sizes)                   # positional writes: no order needed among the ranks
    dist.barrier()                                              # the archive is complete when any rank returns
    return int(all_sizes.sum())


def pack_sharded_set(packer, in_prefixes, out_prefixes, dist, device=None):
    """A SET of libraries as one bin-sharded job: every rank codes its LPT share of EVERY library's bins in one device
    pipeline (so a rank's long streams of all libraries overlap), then the same exchange: ONE all-reduce over the
    concatenated block-size tables, positional writes per library.  Returns the total .cdata bytes."""
    import torch
    tables = packer.shard_pack_set(list(in_prefixes))
    cat = np.concatenate([t[1].astype(np.int64) for t in tables]) if tables else np.zeros(0, dtype=np.int64)
    t = torch.from_numpy(cat)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                    # still the one collective of the path
    all_sizes = t.cpu().numpy().astype(np.uint64)
    at = 0
    for i, (sigs, _) in enumerate(tables):
        packer.shard_write_of(i, out_prefixes[i], all_sizes[at:at + len(sigs)])
        at += len(sigs)
    dist.barrier()
    return int(all_sizes.sum())
