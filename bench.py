#!/usr/bin/env python3
"""bench.py -- fastore_pack compressed MB/s (input FASTQ) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the pack hot path (`fastore_pack e`: read .b*, read-cluster modelling, PPMd /
range-coder entropy coding on the GPU, write .c*) over the workload.  Prints ONE JSON line (rank 0).

Workload (BASELINE.json configs[1]): ONE library of 10 M x 150 bp single-end synthetic reads
(tools/gen_fastq, genome 30 Mbp, seed 8), --lossless, C1 profile, binned by the real reference tools
(oracle/_ref: fastore_bin + 3 x fastore_rebin, untimed, ~3 min on 8 cores).  Its 1 093 standard bins hold 256 ..
47 660 reads, so its quality streams run up to 7.15 M PPMd symbols: the shape that decides the device step.
`--paired` packs ONE library of --reads pairs instead (configs[2] scaled by the stated factor).

--gpus N > 1 (default, "scaling": "weak" -- per-GPU work fixed): the job is a SET of N such libraries (seeds 8 .. 8+N-1,
prepared side by side, one per rank); every rank codes its LPT share (over the .bmeta per-signature totals) of EVERY
library's bins in one device pipeline, the only collective is ONE all-reduce of the concatenated block-size tables (one
u64 per block) over RCCL, every rank writes its blocks at their offsets in the N archives.  `--strong` shards the ONE
library of the N = 1 run instead (its step is bound by single streams, which more GPUs do not shorten: expect a flat
curve); `--replicas` makes every rank pack the whole library into its own archive.

cpu_baseline = the real reference fastore_pack (oracle/_ref) on the same library at -t min(32, cores) (and at
-t1 with --cpu-t1, ~4 min); parity = every block of the product's archive against the reference's block of the same
signature, and the product's block order against the -t1 order (block 0, then ascending signature).
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import time

# the pack context runs one HIP stream per pipeline slice; they need their own hardware queues to overlap
# (the runtime default is 4, shared round-robin) -- must be in the environment before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
REF_GCC = os.path.join(ROOT, "oracle", "_ref", "ref_driver_gcc")
GEN = os.path.join(ROOT, "build", "gen_fastq")
PACK_FLAGS = ["-r", "-f256", "-c10", "-d8", "-w1024", "-W1024"]
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy rate


def sh(cmd, **kw):
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)


def prepare_library(work, name, reads, length, genome, seed, threads, paired=False):
    """FASTQ -> fastore_bin -> 3 x fastore_rebin with the real reference (C1 profile,
    scripts/fastore_compress.sh:146-148,186-209). Cached in `work`."""
    base = os.path.join(work, name)
    binned = base + ".b8"
    pe = ["-z"] if paired else []
    fq = [base + "_1.fastq"] + ([base + "_2.fastq"] if paired else [])
    if not (os.path.exists(binned + ".bmeta") and os.path.exists(base + ".done")):
        sh([GEN, "--reads", str(reads), "--len", str(length), "--genome", str(genome), "--seed", str(seed), "--out", base] + (["--paired"] if paired else []))
        sh([REF_GCC, "bin", "-i" + " ".join(fq), "-o" + base + ".b0", "-t%d" % threads, "-H", "-q0", "-p8", "-s0", "-b256"] + pe)
        prev = base + ".b0"
        for p in (2, 4, 8):
            cur = base + ".b%d" % p
            sh([REF_GCC, "rebin", "-i" + prev, "-o" + cur, "-t%d" % threads, "-r", "-w1024", "-W1024", "-p%d" % p] + pe)
            for e in ("bmeta", "bdna", "bqua", "bhead"):
                if os.path.exists(prev + "." + e):
                    os.remove(prev + "." + e)
            prev = cur
        size = sum(os.path.getsize(f) for f in fq)
        for f in fq:                       # the FASTQ itself is not needed again: only its size enters the metric
            os.remove(f)
        open(base + ".done", "w").write(str(size))
    return binned, int(open(base + ".done").read())


def read_archive(prefix):
    """signature -> block bytes, and the block order, of <prefix>.{cmeta,cdata}"""
    m = open(prefix + ".cmeta", "rb").read()
    foff, _ = struct.unpack_from("<QQ", m, 0)
    n, = struct.unpack_from("<I", m, foff)
    sizes = struct.unpack_from("<%dQ" % n, m, foff + 4)
    sigs = struct.unpack_from("<%dI" % n, m, foff + 4 + 8 * n)
    return sizes, sigs


def same_blocks(ours, ref):
    """every block of `ours` equals the reference block of the same signature; ours is in -t1 order"""
    so, go = read_archive(ours); sr, gr = read_archive(ref)
    if sorted(go) != sorted(gr) or len(go) != len(set(go)):
        return False
    if list(go[1:]) != sorted(go[1:]) or (len(go) > 1 and go[0] < go[-1] and list(go) != sorted(go)):
        return False                       # block 0 (signature 4^p, the largest value) first, then ascending
    off, pos = {}, 0
    for s, g in zip(sr, gr):
        off[g] = (pos, s); pos += s
    with open(ours + ".cdata", "rb") as fo, open(ref + ".cdata", "rb") as fr:
        for s, g in zip(so, go):
            p, rs = off[g]
            if rs != s:
                return False
            fr.seek(p)
            if fo.read(s) != fr.read(s):
                return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads (pairs with --paired) of the ONE library (10 M = configs[1])")
    ap.add_argument("--work", default=os.environ.get("FASTORE_BENCH_DIR", "/tmp/fastore_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-t1", action="store_true", help="also time the reference at -t1 on the same library (~4 min)")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end run of the fastore_pack CLI (process start -> exit)")
    ap.add_argument("--paired", action="store_true", help="ONE paired-end library of --reads pairs: configs[2] scaled, not the default line")
    ap.add_argument("--rehearse", action="store_true", help="N ranks on ONE device over gloo (no RCCL): a dry run of the N > 1 code path on a one-GPU box")
    ap.add_argument("--strong", action="store_true", help="--gpus N: shard the ONE library of the N = 1 run over the ranks (total work fixed)")
    ap.add_argument("--replicas", "--weak", dest="replicas", action="store_true", help="--gpus N: every rank packs the whole library into its own archive")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            local = 0; torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    os.makedirs(args.work, exist_ok=True)
    if rank == 0 and not os.path.exists(GEN):
        subprocess.check_call(["g++", "-O2", "-o", GEN, os.path.join(ROOT, "tools", "gen_fastq.cpp")])
    if not (os.path.exists(REF) and os.path.exists(REF_GCC)):
        raise SystemExit("bench.py needs the reference tools under oracle/_ref (built by __graft_entry__.build()) to bin the synthetic FASTQ")

    L = 150
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    name = ("pe%dk" if args.paired else "se%dk") % (args.reads // 1000)
    cov = 2 if args.paired else 1            # bases per record: the genome is sized for ~50x coverage either way
    prep_s = 0.0
    lib_set = world > 1 and not args.strong and not args.replicas       # the default N > 1 job: N libraries, bin-sharded
    genome = cov * args.reads * L // 50
    if lib_set:
        # rank r prepares library r (seed 8 + r) while the others prepare theirs
        names = [name if r == 0 else "%s_s%d" % (name, 8 + r) for r in range(world)]
        t0 = time.time()
        prepare_library(args.work, names[rank], args.reads, L, genome, 8 + rank, max(2, min(cores // world, 32)), args.paired)
        prep_s = time.time() - t0
        dist.barrier()
        binned_set = [os.path.join(args.work, n + ".b8") for n in names]
        fastq_bytes = sum(int(open(os.path.join(args.work, n + ".done")).read()) for n in names)
        binned = binned_set[0]
    else:
        if rank == 0:
            t0 = time.time()
            binned, fastq_bytes = prepare_library(args.work, name, args.reads, L, genome, 8, min(cores, 32), args.paired)
            prep_s = time.time() - t0
        if world > 1:
            dist.barrier()
        binned = os.path.join(args.work, name + ".b8"); fastq_bytes = int(open(os.path.join(args.work, name + ".done")).read())

    import fastore_amd
    from fastore_amd import shard
    # FASTORE_AMD_LIB: A/B runs of alternative builds of the library (kernel experiments); default = the in-tree build
    alt = os.environ.get("FASTORE_AMD_LIB")
    lib = fastore_amd.load_library(alt) if alt else None
    sharded = world > 1 and not args.replicas
    threads = max(2, cores // world) if world > 1 else 0
    packer = fastore_amd.Packer(device_id=local if world > 1 else 0, lib=lib, host_threads=threads,
                                rank=rank if sharded else 0, world_size=world if sharded else 1)
    out_base = os.path.join(args.work, "out" if sharded else "out_r%d" % rank)
    out = out_base

    # every step writes a NEW archive, as every run of fastore_pack does (overwriting the previous step's 0.45 GB file makes
    # the open wait for its write-back: ~50 ms per step that no real run pays); the extra archives are removed after the timing
    made = []

    # (a long run -- many steps -- must not fill the work directory with 0.45 GB archives: those more than four steps old are
    # removed by a thread of their own while the next step runs; the last one stays for the parity check)
    import threading
    removed = set()

    def drop(prefix):
        for oo in ([prefix] if not lib_set else ["%s_l%d" % (prefix, i) for i in range(world)]):
            for e in (".cdata", ".cmeta"):
                try:
                    os.remove(oo + e)
                except OSError:
                    pass

    def step():
        nonlocal out
        if len(made) > 4 and (rank == 0 or not sharded):
            old = made[len(made) - 5]
            if old not in removed:
                removed.add(old)
                threading.Thread(target=drop, args=(old,), daemon=True).start()
        out = out_base + "_%d" % len(made)
        made.append(out)
        if lib_set:
            shard.pack_sharded_set(packer, binned_set, ["%s_l%d" % (out, i) for i in range(world)], dist, device=None if args.rehearse else torch.device("cuda", local))
        elif sharded:
            shard.pack_sharded(packer, binned, out, dist, device=None if args.rehearse else torch.device("cuda", local))
        else:
            packer.pack_file(binned, out)

    for _ in range(args.warmup):
        step()
    packer.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    st = packer.stats()
    if rank == 0 or not sharded:
        for o in made[:-1]:
            drop(o)
    if world > 1:
        t = torch.tensor([dt], device="cpu" if args.rehearse else "cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        keys = ["algorithmic_bytes", "ppmd_symbols", "host_coded_symbols", "kernel_launches", "encode_kernel_ms", "cdata_bytes", "bins", "records"]
        v = torch.tensor([float(st[k]) for k in keys], device="cpu" if args.rehearse else "cuda", dtype=torch.float64); dist.all_reduce(v)
        tot = dict(zip(keys, v.tolist()))
    else:
        tot = st

    if rank == 0:
        jobs = world if (world > 1 and args.replicas) else 1
        value = fastq_bytes * jobs * args.steps / dt / 1e6
        launches = max(1, int(tot["kernel_launches"]))
        avg_launch_s = tot["encode_kernel_ms"] / 1e3 / launches
        achieved = tot["algorithmic_bytes"] / launches / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC passes cannot run inside this process; the committed summary of the
        # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command is reported per launch
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")
        if os.path.exists(tf) and args.reads == 10_000_000 and not args.paired and world == 1:
            traffic = json.load(open(tf))["hbm_bytes_per_step"] / max(1.0, launches / args.steps)
        sym = max(1.0, float(tot["ppmd_symbols"]))
        res = {
            "metric": "fastore_pack compressed MB/s (input FASTQ)", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak" if (world > 1 and not args.strong) else "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s of %.1f M x %d bp %s synthetic FASTQ (gen_fastq genome %d bp, seed%s), --lossless, C1 profile%s"
                                   % ("ONE library" if not lib_set else "a SET of %d libraries, each" % world, args.reads / 1e6, L, "PE pairs" if args.paired else "SE reads", genome,
                                      " 8" if not lib_set else "s 8..%d" % (7 + world),
                                      "" if not args.paired else " (configs[2] scaled by %g)" % (args.reads / 100e6)),
                       "fastq_bytes": fastq_bytes, "pack_flags": " ".join(PACK_FLAGS),
                       "parallelism": ("1 GPU" if world == 1 else ("%d ranks, each the whole library (replicas)" % world if args.replicas else
                                       ("%d ranks pack disjoint LPT shards of the library's bins; one all-reduce of the block-size table over RCCL; no block bytes cross ranks" % world if args.strong else
                                        "%d ranks, each its LPT share of the bins of all %d libraries in one device pipeline; one all-reduce of the concatenated block-size tables over RCCL; no block bytes cross ranks" % (world, world))))},
            "roofline": {"bound": "hbm", "kernel": "fs_encode_streams", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic,
                         "traffic_unit": "bytes per launch (profiles/r02_hbm_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
                         "avg_launch_ms": round(avg_launch_s * 1e3, 3), "launches": launches,
                         "algorithmic_bytes_per_launch": int(tot["algorithmic_bytes"]) // launches,
                         # the launches of a step overlap (one per pipeline slice, each on its own HIP stream), so a launch's
                         # duration includes the time it shares the GPU; the whole-GPU symbol rate is quoted per step wall time
                         "overlapping_launches_per_step": launches // args.steps,
                         "ppmd_symbols_per_s_whole_job": round(tot["ppmd_symbols"] / dt, 1)},
            "host_coded_symbol_fraction": round(float(tot.get("host_coded_symbols", 0)) / sym, 4),
            # the other kernels of the path (rank 0's context): fs_gather_quality builds the quality streams on the device
            # (HBM-bound: 0.75 B read + 1 B written per score, launch durations from HIP events on the lanes' streams);
            # fs_match_reads does the LZ-window searches of the heaviest bins (duration summed over its launches)
            "other_kernels": {
                "fs_gather_quality": {"ms_per_step": round(st["gather_kernel_ms"] / args.steps, 3), "scores_per_step": st["gather_symbols"] // args.steps,
                                      "achieved_GBps": round(st["gather_bytes"] / max(1e-9, st["gather_kernel_ms"] / 1e3) / 1e9, 1) if st["gather_kernel_ms"] > 0 else None,
                                      "frac_of_hbm_peak": round(st["gather_bytes"] / max(1e-9, st["gather_kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if st["gather_kernel_ms"] > 0 else None},
                "fs_match_reads": {"reads_per_step": st["matcher_reads"] // args.steps, "kernel_ms_per_step": round(st["matcher_kernel_ms"] / args.steps, 1),
                                   "host_wait_ms_per_step_summed_over_threads": round(st["matcher_call_ms"] / args.steps, 1)}},
            "h2d_bytes_per_step": int(st["h2d_bytes"]) // args.steps,
            "stages_ms_per_step_rank0": {k: round(st[k] / args.steps, 1) for k in ("encode_kernel_ms", "assemble_kernel_ms", "frontend_ms", "io_ms", "block0_ms", "total_ms")},
            "archive": {"cdata_bytes": int(tot["cdata_bytes"]) // args.steps, "bins": int(tot["bins"]) // args.steps, "records": int(tot["records"]) // args.steps,
                        "block0_records": st["block0_records"] // args.steps},
            "device": packer.device_name, "host_cores": cores, "prep_s": round(prep_s, 1),
        }
        pe = ["-z"] if args.paired else []
        if not args.no_cli and not lib_set:
            # wall time of the `fastore_pack e` PROCESS (start -> exit: HIP init, arena allocation, reading .b*,
            # writing .c*), page cache warm -- beside the warm in-process number above
            cli = [fastore_amd.PACK_CLI, "e", "-i" + binned, "-o" + os.path.join(args.work, "cli")] + PACK_FLAGS + pe + (["-G%d" % world] if world > 1 else [])
            if world == 1:
                packer.close()             # the process is measured on a device that is otherwise idle, as a user would run it
            runs = []
            for _ in range(3):
                t = time.perf_counter(); rc = subprocess.call(cli, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); tc = time.perf_counter() - t
                runs.append(round(tc, 2))
                if rc != 0:
                    break
            tc = min(runs)
            # (the first processes on a fresh box wait 1-4 s in their first large device allocation while the driver clears
            # memory it has not handed out before -- profiles/r02_mm_alloc_sizes.txt, r02_ll_pool_probe.txt: not the
            # program's time --, hence three runs; the best one is quoted, all three and their median are listed)
            res["cli_end_to_end"] = {"value": round(fastq_bytes / tc / 1e6, 2) if rc == 0 else None, "unit": "MB/s", "seconds": tc, "runs_seconds": runs,
                                     "median_seconds": sorted(runs)[len(runs) // 2], "quoted": "best of the runs", "exit": rc,
                                     "archive_identical_to_the_in_process_one": bool(rc == 0 and world == 1 and all(open(os.path.join(args.work, "cli") + e, "rb").read() == open(out + e, "rb").read() for e in (".cdata", ".cmeta"))),
                                     "command": "fastore_pack e " + " ".join(PACK_FLAGS + pe)}
        if not args.no_cpu_baseline and not lib_set:
            # the reference's multi-threaded pack dead-locks at -t64 (observed here and in the build container), so the
            # all-cores leg uses at most 32 workers, under a timeout, stepping down if it still hangs
            refp = os.path.join(args.work, "ref")
            nt, tn = None, None
            for cand in (32, 16, 8, 4):
                if cand > max(4, cores):
                    continue
                try:
                    t = time.perf_counter()
                    subprocess.run([REF, "pack", "-i" + binned, "-o" + refp, "-t%d" % cand] + PACK_FLAGS + pe, stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, timeout=900, check=True)
                    nt, tn = cand, time.perf_counter() - t
                    break
                except (subprocess.TimeoutExpired, subprocess.CalledProcessError):
                    continue
            if nt is not None:
                res["cpu_baseline"] = {"value": round(fastq_bytes / tn / 1e6, 2), "unit": "MB/s", "cores": min(nt, cores), "kind": "reference",
                                       "sample": "reference fastore_pack e -t%d on the SAME library (whole workload, %.1f MB FASTQ), %d host cores" % (nt, fastq_bytes / 1e6, cores),
                                       "threads": nt, "seconds": round(tn, 2)}
                res["parity"] = {"every_block_bit_identical_to_reference": bool(same_blocks(out, refp)), "block_order": "-t1 (block 0, ascending signature)",
                                 "on": "the whole workload archive (%d blocks)" % len(read_archive(out)[0])}
            if args.cpu_t1:
                t = time.perf_counter(); sh([REF, "pack", "-i" + binned, "-o" + refp + "1", "-t1"] + PACK_FLAGS + pe); t1 = time.perf_counter() - t
                res.setdefault("cpu_baseline", {"unit": "MB/s", "kind": "reference"}).update({"t1_value": round(fastq_bytes / t1 / 1e6, 2), "t1_seconds": round(t1, 2)})
                res.setdefault("parity", {})["cdata_bit_identical_to_reference_t1"] = open(out + ".cdata", "rb").read() == open(refp + "1.cdata", "rb").read()
        print(json.dumps(res), flush=True)
    packer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
