"""Bin-sharded packing of ONE archive over several ranks (one process per GPU).

Bins are independent: rank r packs the standard bins i with i % world == r (rank 0 also the
merged small-bins/N block) into `<out>.part<r>`; the only exchange is an all-gather of the per-block
sizes/signatures (a few bytes per bin -- RCCL on GPUs, gloo in the CPU tests), from which every rank
derives the offset of each of its blocks in the final `.cdata` (block 0 first, then ascending signature:
the reference's -t1 order) and writes them there itself.  No block bytes cross ranks.
"""
import os
import struct


def _read_cmeta(path):
    m = open(path, "rb").read()
    foff, fsize = struct.unpack_from("<QQ", m, 0)
    n, = struct.unpack_from("<I", m, foff)
    sizes = list(struct.unpack_from("<%dQ" % n, m, foff + 4))
    sigs = list(struct.unpack_from("<%dI" % n, m, foff + 4 + 8 * n))
    rest = m[foff + 4 + 12 * n: foff + fsize]        # ArchiveConfig + header field table
    return sizes, sigs, rest


def pack_sharded(packer, in_prefix, out_prefix, dist):
    """packer: fastore_amd.Packer created with rank=dist.get_rank(), world_size=dist.get_world_size()."""
    rank, world = dist.get_rank(), dist.get_world_size()
    packer.pack_file(in_prefix, out_prefix)                      # -> out_prefix.part<rank>.{cdata,cmeta}
    part = "%s.part%d" % (out_prefix, rank)
    sizes, sigs, rest = _read_cmeta(part + ".cmeta")
    gathered = [None] * world
    dist.all_gather_object(gathered, (sizes, sigs))              # the one collective of the path
    # global order: the merged small-bins/N block first (its signature is the out-of-range value 4^p and only
    # rank 0 writes it), then the standard bins in ascending signature order
    entries = [(g, r, i, s) for r, (sz, sg) in enumerate(gathered) for i, (s, g) in enumerate(zip(sz, sg))]
    block0 = [e for e in entries if _raw_signature(e[0]) and e[1] == 0 and e[2] == 0]
    std = sorted(e for e in entries if e not in block0)
    order = block0 + std
    offsets, pos = {}, 0
    for g, r, i, s in order:
        offsets[(r, i)] = pos; pos += s
    total = pos
    if rank == 0:
        with open(out_prefix + ".cdata", "wb") as f:
            f.truncate(total)
    dist.barrier()
    with open(part + ".cdata", "rb") as src, open(out_prefix + ".cdata", "r+b") as dst:
        for i, s in enumerate(sizes):
            dst.seek(offsets[(rank, i)]); dst.write(src.read(s))
    dist.barrier()
    if rank == 0:
        n = len(order)
        body = struct.pack("<I", n) + struct.pack("<%dQ" % n, *[e[3] for e in order]) + struct.pack("<%dI" % n, *[e[0] for e in order]) + rest
        with open(out_prefix + ".cmeta", "wb") as f:
            f.write(struct.pack("<QQ8x", 24, len(body))); f.write(body)
    dist.barrier()
    for ext in (".cdata", ".cmeta"):
        os.remove(part + ext)
    return total


def _raw_signature(sig):
    # the merged small-bins/N block carries the out-of-range signature 4^p (a power of four)
    return sig > 0 and (sig & (sig - 1)) == 0 and (sig.bit_length() - 1) % 2 == 0


