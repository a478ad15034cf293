#!/usr/bin/env python3
"""Compare two fastore archives block by block / stream by stream.

Parses the .cmeta footer (u32 count, u64 sizes[], u32 signatures[]; reference
fastore_pack/ArchiveFile.cpp:111-118) and each LZ block header (fastore_pack/FastqCompressor.cpp:56-69,
684-699) and reports the first differing block and, inside it, the first differing stream.
usage: cdata_diff.py <prefixA> <prefixB> [--all]
"""
import struct, sys

NAMES = ["Flag", "LettersX", "Rev", "HardReads", "LzId", "Shift", "Match", "MatchBinary", "TreeShift", "CMatch", "CShift",
         "CLetters", "Quality", "ReadIdToken", "ReadIdValue", "PE_Flag", "PE_LettersX", "PE_Swap", "PE_Hard", "PE_LzId",
         "PE_Shift", "PE_MatchRLE", "PE_MatchBinary"]

def read_meta(prefix):
    m = open(prefix + ".cmeta", "rb").read()
    foff, fsize = struct.unpack_from("<QQ", m, 0)
    n, = struct.unpack_from("<I", m, foff)
    sizes = struct.unpack_from("<%dQ" % n, m, foff + 4)
    sigs = struct.unpack_from("<%dI" % n, m, foff + 4 + 8 * n)
    conf = m[foff + 4 + 12 * n: foff + 4 + 12 * n + 56]
    rest = m[foff + 4 + 12 * n + 56: foff + fsize]
    return sizes, sigs, conf, rest

def parse_block(b, nsig, has_headers, n_streams, range_coded):
    sig, rec = struct.unpack_from(">IQ", b, 0)
    if sig == nsig:
        return {"sig": sig, "records": rec, "raw": True}
    off = 4 + 8 + 2 + 8 + 8 + 4 + (8 if has_headers else 0)
    work = struct.unpack_from(">%dQ" % n_streams, b, off)
    comp = struct.unpack_from(">%dQ" % n_streams, b, off + 8 * n_streams)
    pos = 42 + 16 * n_streams
    streams = {}
    order = [i for i in range(n_streams) if range_coded(i)] + [i for i in range(n_streams) if not range_coded(i)]
    for i in order:
        streams[i] = b[pos:pos + comp[i]]; pos += comp[i]
    return {"sig": sig, "records": rec, "raw": False, "work": work, "comp": comp, "streams": streams, "header": b[:42 + 16 * n_streams]}

def main():
    a, b = sys.argv[1], sys.argv[2]
    show_all = "--all" in sys.argv
    sa, ga, ca, ra = read_meta(a); sb, gb, cb, rb = read_meta(b)
    print("blocks: %d vs %d" % (len(sa), len(sb)))
    if ga != gb: print("signature lists differ")
    read_type, _, has_headers = ca[0], ca[1], ca[2]
    sig_len = ca[3]; qmethod = ca[16]
    nsig = 1 << (2 * sig_len)
    n_streams = 23 if read_type == 1 else 15
    rc = lambda i: i in (1, 2, 7, 11, 13, 14, 15, 16, 22) or (i == 12 and qmethod != 0)
    if ca[:3] != cb[:3] or ca[3:11] != cb[3:11] or ca[16:18] != cb[16:18]: print("archive config differs")
    if ra != rb: print("cmeta tail (header field table) differs")
    da = open(a + ".cdata", "rb"); db = open(b + ".cdata", "rb")
    ndiff = 0
    for i in range(min(len(sa), len(sb))):
        xa = da.read(sa[i]); xb = db.read(sb[i])
        if xa == xb: continue
        ndiff += 1
        if ndiff > 1 and not show_all: continue
        print("block %d (sig %d) differs: size %d vs %d" % (i, ga[i], sa[i], sb[i]))
        pa = parse_block(xa, nsig, has_headers, n_streams, rc); pb = parse_block(xb, nsig, has_headers, n_streams, rc)
        if pa["raw"] or pb["raw"]:
            print("  raw block; header A", xa[:74].hex()); print("  raw block; header B", xb[:74].hex()); continue
        print("  records %d vs %d" % (pa["records"], pb["records"]))
        if pa["header"][:34] != pb["header"][:34]: print("  fixed header differs:", pa["header"][:42].hex(), pb["header"][:42].hex())
        for s in range(n_streams):
            if pa["work"][s] != pb["work"][s] or pa["comp"][s] != pb["comp"][s] or pa["streams"][s] != pb["streams"][s]:
                print("  stream %2d %-12s work %d vs %d  comp %d vs %d  %s" % (s, NAMES[s], pa["work"][s], pb["work"][s], pa["comp"][s], pb["comp"][s],
                      "bytes differ" if pa["streams"][s] != pb["streams"][s] else ""))
    print("differing blocks: %d of %d" % (ndiff, len(sa)))
    return 1 if (ndiff or len(sa) != len(sb)) else 0

if __name__ == "__main__":
    sys.exit(main())
