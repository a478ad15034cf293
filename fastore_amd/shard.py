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


def _steal_enabled():
    v = os.environ.get("FS_STEAL", "")
    return v not in ("", "0")


def _job_key(packer, dist, device=None):
    """The work-stealing tail of the split (packer.cpp: StealCounter; off unless FS_STEAL=1, which must then be set on EVERY
    rank): the ranks claim the lightest bins from one counter in /dev/shm, whose name must be the job's own and whose file is
    the NODE's.  Once per packer (control plane -- the data path's collective stays the one all-reduce of the size table per
    pack): rank 0 draws the key (from the launcher's FS_STEAL_KEY if it set one, else at random) and broadcasts it -- every rank
    takes part whatever its own environment says, so the ranks cannot disagree about the collective --, and the ranks compare
    host names: a job over several nodes deals every bin up front (a counter per node would hand every node the whole tail).
    The key lives on the packer, not in the process environment: another job in this process draws its own."""
    if not _steal_enabled():
        return
    key = getattr(packer, "_steal_key", None)
    if key is None:
        import hashlib
        import socket
        import torch
        mine = os.environ.get("FS_STEAL_KEY")
        seed = int.from_bytes(hashlib.sha256(mine.encode()).digest()[:7], "little") if mine else int.from_bytes(os.urandom(7), "little")
        t = torch.tensor([seed], dtype=torch.int64)       # (a plain tensor: the same call under RCCL and gloo)
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=0)
        hosts = [None] * dist.get_world_size()
        dist.all_gather_object(hosts, socket.gethostname())
        key = "%x" % int(t.item()) if len(set(hosts)) == 1 else ""
        packer._steal_key = key
    if key:
        os.environ["FS_STEAL_KEY"] = key; os.environ["FS_STEAL_ONE_NODE"] = "1"
    else:
        os.environ["FS_STEAL"] = "0"                      # several nodes: no tail for this process's packs


def pack_sharded(packer, in_prefix, out_prefix, dist, device=None):
    """packer: fastore_amd.Packer created with rank=dist.get_rank(), world_size=dist.get_world_size().
    device: torch device of the collective's tensor (the rank's GPU under nccl = RCCL; None = CPU for gloo)."""
    import torch
    _job_key(packer, dist, device)
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
    _job_key(packer, dist, device)
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
