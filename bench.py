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

--gpus N > 1 (default, `value`, "scaling": "strong" -- total work fixed): the ONE library of the N = 1 run, bin-sharded over the N
ranks as BASELINE.json's configs[3] / configs[4] ask (every rank codes its LPT share, over the .bmeta per-signature totals, of the
library's bins; the only collective is ONE all-reduce of the block-size table, one u64 per block, over RCCL; every rank writes its
blocks at their offsets in the one archive).  Its step is bound by single streams, which more GPUs do not shorten: reported as
measured.  Beside it, key `weak_set` ("scaling": "weak" -- per-GPU work fixed): a SET of N such libraries (seeds 8 .. 8+N-1,
prepared side by side, one per rank), every rank its share of EVERY library's bins in one device pipeline.  `--strong`: the
headline alone; `--weak-set`: the set as the headline (rounds 3-4's line); `--replicas`: every rank packs the whole library.

cpu_baseline = the real reference fastore_pack (oracle/_ref) on the same library at -t min(32, cores) (and at
-t1 with --cpu-t1, ~4 min); parity = every block of the product's archive against the reference's block of the same
signature, and the product's block order against the -t1 order (block 0, then ascending signature).

Two regimes, both printed and labelled (SURVEY 8(d) defines the metric on the PROCESS): `value` / `ms_per_step` = the K timed
warm steps of the contract (context made, file to file); `process` = the `fastore_pack e` process from start to exit, median of
three runs.  `speedup` holds both ratios against the reference PROCESS; `speedup_vs_cpu_baseline` is the like-with-like one,
process against process.  N = 1 legs beside the headline: `pe` (configs[2] scaled to what the driver's window holds, 15 M pairs by
default), `reduced` (configs[3]'s mode) and `lossy` (QVZ) on the headline's reads.  All legs' libraries are prepared side by side
before the first timed step.
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import threading
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


# quality / read-id mode of the library = what scripts/fastore_compress.sh:135-140 hands to fastore_bin (--lossless / --reduced / --lossy);
# the pack stage's own flags do not change with it
QUALITY_MODES = {"lossless": ["-H", "-q0"], "reduced": ["-H", "-C", "-q2"], "lossy": ["-H", "-C", "-q3"]}
QUALITY = "lossless"


def sh(cmd, **kw):
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)


_T0 = float(os.environ.get("FS_BENCH_T0", "0")) or time.time()       # (a leg's child process counts on from the parent's start)


def say(msg):
    """progress on stderr (the one JSON line on stdout stays alone): a run of several minutes that says nothing looks hung from outside"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1f s] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def prepare_library(work, name, reads, length, genome, seed, threads, paired=False, quality=None):
    """FASTQ -> fastore_bin -> 3 x fastore_rebin with the real reference (C1 profile,
    scripts/fastore_compress.sh:146-148,186-209). Cached in `work`."""
    quality = quality or QUALITY
    base = os.path.join(work, name)
    binned = base + ".b8"
    pe = ["-z"] if paired else []
    fq = [base + "_1.fastq"] + ([base + "_2.fastq"] if paired else [])
    if not (os.path.exists(binned + ".bmeta") and os.path.exists(base + ".done")):
        say("preparing library %s (%d %s, --%s): gen_fastq, fastore_bin, 3 x fastore_rebin at -t%d" % (name, reads, "pairs" if paired else "reads", quality, threads))
        sh([GEN, "--reads", str(reads), "--len", str(length), "--genome", str(genome), "--seed", str(seed), "--out", base] + (["--paired"] if paired else []))
        sh([REF_GCC, "bin", "-i" + " ".join(fq), "-o" + base + ".b0", "-t%d" % threads] + QUALITY_MODES[quality] + ["-p8", "-s0", "-b256"] + pe)
        prev = base + ".b0"
        for p in (2, 4, 8):
            cur = base + ".b%d" % p
            sh([REF_GCC, "rebin", "-i" + prev, "-o" + cur, "-t%d" % threads, "-r", "-w1024", "-W1024", "-p%d" % p] + pe)
            for e in ("bmeta", "bdna", "bqua", "bhead"):
                if os.path.exists(prev + "." + e):
                    os.remove(prev + "." + e)
            prev = cur
            say("library %s: rebin -p%d done" % (name, p))
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


def differing_blocks(ours, ref):
    """(signature, our size, reference size) of the blocks that are not the reference's; None if the block tables do not even match"""
    so, go = read_archive(ours); sr, gr = read_archive(ref)
    if sorted(go) != sorted(gr):
        return None
    off, pos = {}, 0
    for s, g in zip(sr, gr):
        off[g] = (pos, s); pos += s
    bad = []
    with open(ours + ".cdata", "rb") as fo, open(ref + ".cdata", "rb") as fr:
        for s, g in zip(so, go):
            p, rs = off[g]
            fr.seek(p)
            if fo.read(s) != fr.read(rs):
                bad.append((int(g), int(s), int(rs)))
    return bad


def roofline_of(tot, steps, traffic=None):
    """algorithmic bytes per launch of the dominant kernel / its average launch duration (HIP events on the
    lanes' own streams, summed by the library) against the HBM peak."""
    launches = max(1, int(tot["kernel_launches"]))
    avg_launch_s = tot["encode_kernel_ms"] / 1e3 / launches
    achieved = tot["algorithmic_bytes"] / launches / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    return {"bound": "hbm", "kernel": "fs_encode_streams", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic,
            "traffic_unit": "bytes per launch (profiles/*hbm_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
            "avg_launch_ms": round(avg_launch_s * 1e3, 3), "launches": launches,
            "algorithmic_bytes_per_launch": int(tot["algorithmic_bytes"]) // launches,
            # the launches of a step overlap (one per pipeline slice, each on its own HIP stream), so a launch's
            # duration includes the time it shares the GPU; the whole-GPU symbol rate is quoted per step wall time
            "overlapping_launches_per_step": launches // max(1, steps)}


def reference_pack(binned, refp, cores, pe, sweep=False):
    """the real reference fastore_pack on the same library.  Its multi-threaded pack dead-locks at -t64 (observed here and
    in the build container), so the all-cores leg uses at most 32 workers, under a timeout, stepping down if it still hangs.
    sweep: also time 16 and 48 workers once (is -t32 the reference's best on this host?)."""
    nt, tn = None, None
    for cand in (32, 16, 8, 4):
        if cand > max(4, cores):
            continue
        try:
            t = time.perf_counter()
            subprocess.run([REF, "pack", "-i" + binned, "-o" + refp, "-t%d" % cand] + PACK_FLAGS + pe, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=1500, check=True)
            nt, tn = cand, time.perf_counter() - t
            break
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError):
            continue
    others = {}
    if sweep and nt is not None:
        for cand in (16, 48):
            if cand == nt or cand > cores:
                continue
            try:
                t = time.perf_counter()
                subprocess.run([REF, "pack", "-i" + binned, "-o" + refp + "_sweep", "-t%d" % cand] + PACK_FLAGS + pe, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=600, check=True)
                others["t%d_seconds" % cand] = round(time.perf_counter() - t, 2)
            except (subprocess.TimeoutExpired, subprocess.CalledProcessError):
                others["t%d_seconds" % cand] = None
        for e in (".cdata", ".cmeta"):
            try:
                os.remove(refp + "_sweep" + e)
            except OSError:
                pass
    return nt, tn, others


def cli_phase(fastore_amd, args, work, name, reads, paired, genome, cores, cli_runs, quality=None):
    """wall time of the `fastore_pack e` PROCESS (start -> exit: HIP init, arena allocation, reading .b*, writing .c*), page
    cache warm.  Run for every leg BEFORE this process touches the device: a user's fastore_pack does not share the GPU with a second
    process that holds sixteen hardware queues of its own -- with one, the child's sixteen are more than the device maps at once, launches
    of the child wait for queues that never come free and the process stands still (found in round 4: profiles/r04_cli_two_processes.txt).
    The archive of the last run is kept for the comparison with the in-process one."""
    quality = quality or QUALITY
    binned, fastq_bytes = prepare_library(work, name, reads, 150, genome, 8, min(cores, 32), paired, quality)
    pe = ["-z"] if paired else []
    cli_out = os.path.join(work, "cli_" + name)
    cli = [fastore_amd.PACK_CLI, "e", "-i" + binned, "-o" + cli_out] + PACK_FLAGS + pe
    runs, rc = [], 0
    say("leg %s: the fastore_pack e process, %d runs" % (name, cli_runs))
    for _ in range(cli_runs):
        t = time.perf_counter()
        try:
            # (FS_BENCH_CLI_LOG=<file>: the process's stderr -- its FS_TRACE timeline, a watchdog's report -- is kept)
            with open(os.environ["FS_BENCH_CLI_LOG"], "ab") if os.environ.get("FS_BENCH_CLI_LOG") else open(os.devnull, "wb") as errf:
                rc = subprocess.call(cli, stdout=subprocess.DEVNULL, stderr=errf, timeout=int(os.environ.get("FS_BENCH_CLI_TIMEOUT", "90")))
        except subprocess.TimeoutExpired:      # (a process that does not end is reported as such, not waited for)
            rc = -9
        tc = time.perf_counter() - t
        runs.append(round(tc, 2))
        if rc != 0:
            break
    med = sorted(runs)[len(runs) // 2]
    # (the first processes on a fresh box wait 1-4 s in their first large device allocation while the driver clears
    # memory it has not handed out before -- profiles/r02_mm_alloc_sizes.txt: not the program's time --, hence
    # several runs; the MEDIAN is quoted, all runs are listed)
    return {"value": round(fastq_bytes / med / 1e6, 2) if rc == 0 else None, "unit": "MB/s", "seconds": med, "runs_seconds": runs,
            "what": "SURVEY 8(d)'s metric: the fastore_pack e PROCESS, start -> exit (HIP init, arena allocation, .b* in, .c* out), page cache warm; run before this process opened the device",
            "best_seconds": min(runs), "quoted": "median of the runs", "exit": rc, "command": "fastore_pack e " + " ".join(PACK_FLAGS + pe), "_archive": cli_out}


def one_library_leg(fastore_amd, torch, args, work, name, reads, paired, genome, steps, warmup, cores, lib, cli, traffic_file=None, sweep=False, quality=None):
    """ONE library on ONE GPU: K timed pack steps (file to file), then the CLI process, the reference on the same library,
    and the block-for-block comparison.  Returns the leg's result dict."""
    L = 150
    quality = quality or QUALITY
    t0 = time.time()
    binned, fastq_bytes = prepare_library(work, name, reads, L, genome, 8, min(cores, 32), paired, quality)
    prep_s = time.time() - t0
    packer = fastore_amd.Packer(device_id=0, lib=lib, host_threads=0)
    out_base = os.path.join(work, "out_" + name)
    made, removed = [], set()

    def drop(prefix):
        for e in (".cdata", ".cmeta"):
            try:
                os.remove(prefix + e)
            except OSError:
                pass

    def step():
        # every step writes a NEW archive, as every run of fastore_pack does (overwriting the previous step's file makes the
        # open wait for its write-back: ~50 ms per step that no real run pays); archives more than four steps old are
        # removed by a thread of their own while the next step runs; the last one stays for the parity check
        if len(made) > 4:
            old = made[len(made) - 5]
            if old not in removed:
                removed.add(old)
                threading.Thread(target=drop, args=(old,), daemon=True).start()
        o = out_base + "_%d" % len(made)
        made.append(o)
        packer.pack_file(binned, o)

    say("leg %s: %d warm-up + %d timed steps" % (name, warmup, steps))
    for _ in range(warmup):
        step()
    packer.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    each = []
    for _ in range(steps):
        t1 = time.perf_counter(); step(); each.append(round((time.perf_counter() - t1) * 1e3, 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    say("leg %s: %.1f ms per step" % (name, dt / steps * 1e3))
    st = packer.stats()
    out = made[-1]
    for o in made[:-1]:
        drop(o)
    traffic = None
    if traffic_file and os.path.exists(traffic_file):
        # HBM bytes need the PMC counters, i.e. rocprofv3 around the process: they come from the committed counter passes of this
        # very command (tools/pmc_passes.sh -> tools/hbm_traffic.py; the file is named in the line), not from this run
        traffic = json.load(open(traffic_file))["hbm_bytes_per_step"] / max(1.0, int(st["kernel_launches"]) / steps)
    sym = max(1.0, float(st["ppmd_symbols"]))
    rf = roofline_of(st, steps, traffic)
    if traffic is not None:
        rf["traffic_source"] = os.path.relpath(traffic_file, ROOT)
    rf["ppmd_symbols_per_s_whole_job"] = round(st["ppmd_symbols"] / dt, 1)
    res = {
        "value": round(fastq_bytes * steps / dt / 1e6, 2), "unit": "MB/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2), "each_step_ms": each,
        "config": {"workload": "ONE library of %.1f M x %d bp %s synthetic FASTQ (gen_fastq genome %d bp, seed 8), --%s, C1 profile%s"
                               % (reads / 1e6, L, "PE pairs" if paired else "SE reads", genome, quality, "" if not paired else " (configs[2] scaled by %g)" % (reads / 100e6)),
                   "fastq_bytes": fastq_bytes, "pack_flags": " ".join(PACK_FLAGS + (["-z"] if paired else [])), "parallelism": "1 GPU"},
        "roofline": rf,
        "host_coded_symbol_fraction": round(float(st.get("host_coded_symbols", 0)) / sym, 4),
        # the other kernels of the path: fs_gather_quality builds the quality streams on the device (HBM-bound: 0.75 B
        # read + 1 B written per score, launch durations from HIP events on the lanes' streams); fs_match_reads does the
        # LZ-window searches (duration summed over its launches)
        "other_kernels": {
            "fs_gather_quality": {"ms_per_step": round(st["gather_kernel_ms"] / steps, 3), "scores_per_step": st["gather_symbols"] // steps,
                                  "achieved_GBps": round(st["gather_bytes"] / max(1e-9, st["gather_kernel_ms"] / 1e3) / 1e9, 1) if st["gather_kernel_ms"] > 0 else None,
                                  "frac_of_hbm_peak": round(st["gather_bytes"] / max(1e-9, st["gather_kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if st["gather_kernel_ms"] > 0 else None},
            "fs_match_reads": {"reads_per_step": st["matcher_reads"] // steps, "kernel_ms_per_step": round(st["matcher_kernel_ms"] / steps, 1),
                               "host_wait_ms_per_step_summed_over_threads": round(st["matcher_call_ms"] / steps, 1),
                               # its bases: unpacked on the device from the bin's .bdna bytes (fs_unpack_planes), or ASCII from the host
                               "reads_unpacked_on_device_per_step": st["matcher_unpacked_reads"] // steps, "bases_h2d_bytes_per_step": st["matcher_bases_h2d_bytes"] // steps},
            # the mate searches of paired-end bins that ran on the device (0: the host's search did them)
            "fs_match_mates": {"pairs_per_step": st["mate_pairs"] // steps, "kernel_ms_per_step": round(st["mate_kernel_ms"] / steps, 1),
                               "call_ms_per_step_summed_over_callers": round(st["mate_call_ms"] / steps, 1)}},
        "h2d_bytes_per_step": int(st["h2d_bytes"]) // steps,
        "stages_ms_per_step": dict({k: round(st[k] / steps, 1) for k in ("encode_kernel_ms", "assemble_kernel_ms", "frontend_ms", "io_ms", "total_ms")},
                                   block0_ms=round(st["block0_ms"], 1)),      # block0_ms is the longest single step's (a max in the library), not a sum
        "archive": {"cdata_bytes": int(st["cdata_bytes"]) // steps, "bins": int(st["bins"]) // steps, "records": int(st["records"]) // steps,
                    "block0_records": st["block0_records"] // steps},
        "prep_s": round(prep_s, 1),
    }
    device_name = packer.device_name
    pe = ["-z"] if paired else []
    packer.close()
    if cli:       # (cli_phase's result: the process was timed before this one opened the device; its archive against the in-process one)
        cli_out = cli.pop("_archive")
        cli["archive_identical_to_the_in_process_one"] = bool(cli["exit"] == 0 and all(open(cli_out + e, "rb").read() == open(out + e, "rb").read() for e in (".cdata", ".cmeta")))
        res["process"] = cli
        drop(cli_out)
    if not args.no_cpu_baseline:
        refp = os.path.join(work, "ref_" + name)
        say("leg %s: the reference's fastore_pack on the same library" % name)
        nt, tn, others = reference_pack(binned, refp, cores, pe, sweep)
        say("leg %s: reference done (%s s at -t%s); comparing every block" % (name, "%.1f" % tn if tn else "-", nt))
        if nt is not None:
            res["cpu_baseline"] = {"value": round(fastq_bytes / tn / 1e6, 2), "unit": "MB/s", "cores": min(nt, cores), "kind": "reference",
                                   "sample": "reference fastore_pack e -t%d on the SAME library (whole workload, %.1f MB FASTQ), %d host cores" % (nt, fastq_bytes / 1e6, cores),
                                   "threads": nt, "seconds": round(tn, 2)}
            if others:
                res["cpu_baseline"]["thread_sweep"] = others
            # like with like: process against process; the warm step against the reference's process is the other regime, labelled
            ref_v = res["cpu_baseline"]["value"]
            res["speedup"] = {"process_vs_reference_process": round(res["process"]["value"] / ref_v, 2) if res.get("process", {}).get("value") else None,
                              "warm_step_vs_reference_process": round(res["value"] / ref_v, 2)}
            res["speedup_vs_cpu_baseline"] = res["speedup"]["process_vs_reference_process"] or res["speedup"]["warm_step_vs_reference_process"]
            res["speedup_vs_cpu_baseline_regime"] = "process vs process" if res["speedup"]["process_vs_reference_process"] else "warm step vs reference process (no CLI run)"
            same = bool(same_blocks(out, refp))
            res["parity"] = {"every_block_bit_identical_to_reference": same, "block_order": "-t1 (block 0, ascending signature)",
                             "on": "the whole workload archive (%d blocks)" % len(read_archive(out)[0])}
            if not same:
                # Which blocks, and does the reference agree with ITSELF?  (Round 5: one 25 M-pair leg in some thirty reported a mismatch that a
                # second pack of the same library by the same build, and a second run of the reference, did not have: tools/pe_parity_debug.sh.)
                # The verdict above stays what the first comparison said; what a second run of the reference gives is reported beside it.
                bad = differing_blocks(out, refp)
                res["parity"]["differing_blocks"] = None if bad is None else {"count": len(bad), "first": bad[:4]}
                try:
                    nt2, _, _ = reference_pack(binned, refp + "_again", cores, pe, False)
                    if nt2 is not None:
                        res["parity"]["against_a_second_run_of_the_reference"] = bool(same_blocks(out, refp + "_again"))
                        again = differing_blocks(refp + "_again", refp)
                        res["parity"]["the_reference_runs_differ_in_blocks"] = None if again is None else len(again)
                finally:
                    drop(refp + "_again")
        if args.cpu_t1:
            t = time.perf_counter(); sh([REF, "pack", "-i" + binned, "-o" + refp + "1", "-t1"] + PACK_FLAGS + pe); t1 = time.perf_counter() - t
            res.setdefault("cpu_baseline", {"unit": "MB/s", "kind": "reference"}).update({"t1_value": round(fastq_bytes / t1 / 1e6, 2), "t1_seconds": round(t1, 2)})
            res.setdefault("parity", {})["cdata_bit_identical_to_reference_t1"] = open(out + ".cdata", "rb").read() == open(refp + "1.cdata", "rb").read()
            drop(refp + "1")
        drop(refp)
    drop(out)
    return res, device_name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads (pairs with --paired) of the ONE library (10 M = configs[1])")
    ap.add_argument("--work", default=os.environ.get("FASTORE_BENCH_DIR", "/tmp/fastore_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-t1", action="store_true", help="also time the reference at -t1 on the same library (~4 min)")
    ap.add_argument("--cpu-sweep", action="store_true", help="also time the reference at -t16 and -t48 once")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end run of the fastore_pack CLI (process start -> exit)")
    ap.add_argument("--paired", action="store_true", help="the headline leg packs ONE paired-end library of --reads pairs (configs[2] scaled)")
    ap.add_argument("--pe-reads", type=int, default=25_000_000, help="pairs of the paired-end leg of the N = 1 line (configs[2] scaled to what the run's window holds)")
    ap.add_argument("--weak-set", action="store_true", help="--gpus N: the SET of N libraries is the headline (`value`), the ONE-library line the side key")
    ap.add_argument("--no-pe", action="store_true", help="N = 1: the headline leg only (skips the paired-end leg and the --reduced leg)")
    ap.add_argument("--no-reduced", action="store_true", help="N = 1: skip the --reduced leg (the same reads with 8-bin quality scores)")
    ap.add_argument("--no-lossy", action="store_true", help="N = 1: skip the --lossy leg (the same reads with QVZ-coded quality scores)")
    ap.add_argument("--rehearse", action="store_true", help="N ranks on ONE device over gloo (no RCCL): a dry run of the N > 1 code path on a one-GPU box")
    ap.add_argument("--strong", action="store_true", help="--gpus N: ONLY the strong line (the ONE library of the N = 1 run sharded over the ranks)")
    ap.add_argument("--replicas", "--weak", dest="replicas", action="store_true", help="--gpus N: every rank packs the whole library into its own archive")
    ap.add_argument("--quality", choices=sorted(QUALITY_MODES), default="lossless",
                    help="mode of the library as scripts/fastore_compress.sh names it: lossless (-q0, the BASELINE metric's), reduced (8-bin scores, -q2 -C: configs[3]'s), lossy (QVZ, -q3 -C)")
    ap.add_argument("--in-process", action="store_true", help="N = 1: measure the legs in THIS process instead of in a process each (profiler runs: a process that has opened the device must not start another)")
    ap.add_argument("--one-leg", default=None, help=argparse.SUPPRESS)      # (internal: this process measures ONE leg, described by the JSON file named, and prints its result)
    args = ap.parse_args()
    global QUALITY
    QUALITY = args.quality

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        # several ranks on ONE device: together they must not ask for more hardware queues than the device maps (profiles/r04_cli_two_processes.txt)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(2, 16 // max(1, world))))
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
    name = ("pe%dk" if args.paired else "se%dk") % (args.reads // 1000) + ("" if QUALITY == "lossless" else "_" + QUALITY)
    cov = 2 if args.paired else 1            # bases per record: the genome is sized for ~50x coverage either way
    genome = cov * args.reads * L // 50

    import fastore_amd
    from fastore_amd import shard
    # FASTORE_AMD_LIB: A/B runs of alternative builds of the library (kernel experiments); default = the in-tree build
    alt = os.environ.get("FASTORE_AMD_LIB")
    lib = fastore_amd.load_library(alt) if alt else None

    if args.one_leg:
        spec = json.load(open(args.one_leg))
        leg, dev = one_library_leg(fastore_amd, torch, args, args.work, spec["name"], spec["reads"], spec["paired"], spec["genome"], spec["steps"], spec["warmup"], cores, lib,
                                   spec.get("cli"), spec.get("traffic_file"), spec.get("sweep", False), quality=spec["quality"])
        print(json.dumps({"leg": leg, "device": dev}), flush=True)
        return

    if world == 1:
        # ---- the N = 1 line: the headline leg (configs[1], or --paired) and, beside it, the paired-end, --reduced and --lossy legs ----
        if args.gpus != 1:
            raise SystemExit("bench.py --gpus %d needs %d ranks (python -m torch.distributed.run --nproc-per-node %d ...); WORLD_SIZE is 1" % (args.gpus, args.gpus, args.gpus))
        pk = args.pe_reads
        legs = [("main", name, args.reads, args.paired, genome, QUALITY)]
        if not args.paired and not args.no_pe:
            legs.append(("pe", "pe%dk" % (pk // 1000) + ("" if QUALITY == "lossless" else "_" + QUALITY), pk, True, 2 * pk * L // 50, QUALITY))
            if QUALITY == "lossless" and not args.no_reduced:
                legs.append(("reduced", name + "_reduced", args.reads, False, genome, "reduced"))
            if QUALITY == "lossless" and not args.no_lossy:
                legs.append(("lossy", name + "_lossy", args.reads, False, genome, "lossy"))
        # every leg's library is generated and binned by the reference's tools BEFORE the first timed step, side by side (the
        # stages are untimed, but one after the other they were half of the run's wall time)
        errs = {}

        def prep(leg, workers):
            try:
                prepare_library(args.work, leg[1], leg[2], L, leg[4], 8, workers, leg[3], leg[5])
            except Exception as e:      # noqa: BLE001
                errs[leg[0]] = "%s: %s" % (type(e).__name__, e)
        t0 = time.time()
        # the paired-end library is the long pole (25 M pairs: 16 GB of FASTQ through four stages of the reference's tools): it starts
        # first and keeps 32 workers (the tools' limit is 64, their best here 32); the single-end libraries share the other cores
        big = [leg for leg in legs if leg[3]]; small = [leg for leg in legs if not leg[3]]
        per_small = max(4, min(32, (cores - (32 if big else 0)) // max(1, len(small)))) if cores >= 64 else max(4, min(32, cores // max(1, len(legs))))
        th = [threading.Thread(target=prep, args=(leg, min(32, cores))) for leg in big] + [threading.Thread(target=prep, args=(leg, per_small)) for leg in small]
        for t in th:
            t.start()
        for t in th:
            t.join()
        prep_all_s = time.time() - t0
        if "main" in errs:
            raise SystemExit("could not prepare the headline library: " + errs["main"])
        # HBM bytes per step of every leg: from the committed counter passes of this very command (tools/pmc_passes.sh -> tools/hbm_traffic.py),
        # newest round first; a leg whose passes were not taken says null
        def traffic_of(key):
            if not (args.reads == 10_000_000 and not args.paired and QUALITY == "lossless"):
                return None
            for rnd in ("r05", "r04"):
                f = os.path.join(ROOT, "profiles", "%s_hbm_traffic%s.json" % (rnd, "" if key == "main" else "_" + key))
                if os.path.exists(f) and (key == "main" or args.pe_reads == json.load(open(f)).get("pe_reads", args.pe_reads)):
                    return f
            return None
        traffic_file = traffic_of("main")
        # every leg's fastore_pack e PROCESS first, while this process has not opened the device (cli_phase says why)
        clis = {}
        if not args.no_cli:
            for key, lname, lreads, lpaired, lgenome, lq in legs:
                if key not in errs:
                    try:
                        clis[key] = cli_phase(fastore_amd, args, args.work, lname, lreads, lpaired, lgenome, cores, 3, quality=lq)      # (three runs a leg: one process in ten waits 1-3 s in its arena allocation while the driver takes back what the process before it held -- the median holds)
                    except Exception as e:      # noqa: BLE001
                        say("leg %s: the process could not be timed: %s" % (lname, e))
        # Every leg is measured by a process of its own, one after the other; this process never opens the device.  (Two reasons: a leg that
        # stands still is ended at its time limit and reported in its place instead of taking the line with it; and a process that HAS
        # packed keeps its hardware queues, which a second process on the same device must not meet -- cli_phase.)
        def leg_process(key, lname, lreads, lpaired, lgenome, lq, steps, warmup, limit_s, traffic=None, sweep=False):
            if args.in_process:
                return one_library_leg(fastore_amd, torch, args, args.work, lname, lreads, lpaired, lgenome, steps, warmup, cores, lib, clis.get(key), traffic, sweep, quality=lq)
            spec = {"name": lname, "reads": lreads, "paired": lpaired, "genome": lgenome, "quality": lq, "steps": steps, "warmup": warmup, "cli": clis.get(key),
                    "traffic_file": traffic, "sweep": sweep}
            sf = os.path.join(args.work, "leg_%s.json" % key)
            json.dump(spec, open(sf, "w"))
            env = dict(os.environ, FS_BENCH_T0=repr(_T0))
            # (a session of its own: at its time limit the leg goes with everything it started -- the reference's pack is its grandchild)
            proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--one-leg", sf], stdout=subprocess.PIPE, env=env, start_new_session=True)
            try:
                stdout, _ = proc.communicate(timeout=limit_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise RuntimeError("the leg's process did not end within %d s and was stopped" % limit_s)
            if proc.returncode != 0:
                raise RuntimeError("the leg's process ended with code %d" % proc.returncode)
            out = json.loads(stdout.decode().strip().splitlines()[-1])
            return out["leg"], out["device"]
        ref_s = 0 if args.no_cpu_baseline else 120 + (240 if args.cpu_t1 else 0) + (200 if args.cpu_sweep else 0)
        budget_end = _T0 + float(os.environ.get("FS_BENCH_BUDGET_S", "560"))
        main_limit = 240 + 12 * (args.steps + args.warmup) * max(1, args.reads // 10_000_000 * (3 if args.paired else 1)) + ref_s
        try:
            leg, dev = leg_process("main", name, args.reads, args.paired, genome, QUALITY, args.steps, args.warmup, int(max(60.0, min(main_limit, budget_end - time.time() - 5.0))),
                                   traffic_file, args.cpu_sweep)
        except Exception as e:          # noqa: BLE001  (the line is printed in any case: what was measured before stays in it)
            leg, dev = {"value": None, "ms_per_step": None, "error": "%s: %s" % (type(e).__name__, e), "process": {k: v for k, v in clis.get("main", {}).items() if k != "_archive"} or None}, None

        res = {"metric": "fastore_pack compressed MB/s (input FASTQ)", "value": leg["value"], "unit": "MB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "value_is": "the K timed warm steps (file to file, context made); `process` holds SURVEY 8(d)'s process start -> exit, `speedup` both ratios"}
        # (the keys the line is read for first: the process, both ratios, the roofline, the reference, parity -- then the rest)
        for k in ("config", "process", "speedup", "speedup_vs_cpu_baseline", "speedup_vs_cpu_baseline_regime", "roofline", "cpu_baseline", "parity"):
            if k in leg:
                res[k] = leg[k]
        for k, v in leg.items():
            if k not in res:
                res[k] = v
        if "stages_ms_per_step" in res:
            res["stages_ms_per_step_rank0"] = res.pop("stages_ms_per_step")
        res["device"] = dev; res["host_cores"] = cores; res["prep_all_legs_s"] = round(prep_all_s, 1)
        for key, lname, lreads, lpaired, lgenome, lq in legs[1:]:
            # (a leg beside the headline must never cost the line itself: what goes wrong in it is reported in its place)
            if key in errs:
                res[key] = {"error": errs[key]}
                continue
            ls, lw = max(1, min(args.steps, 3)), 1
            try:
                # (the whole run keeps inside FS_BENCH_BUDGET_S, 560 s by default -- the driver gives the command ten minutes --: a leg gets what is
                # left of it at most, and one that could not even start says so; the line itself is printed in any case)
                left = budget_end - time.time() - 5.0
                if left < 45.0:
                    raise RuntimeError("not started: the run's time budget was used up")
                lleg, _ = leg_process(key, lname, lreads, lpaired, lgenome, lq, ls, lw, int(min((420 if key == "pe" else 240) + ref_s, left)), traffic_of(key))
            except Exception as e:          # noqa: BLE001
                lleg = {"error": "%s: %s" % (type(e).__name__, e)}
            res[key] = lleg
        print(json.dumps(res), flush=True)
        return

    # ---- N > 1 ----
    if args.gpus != world or dist.get_world_size() != world:
        raise SystemExit("bench.py --gpus %d was started with %d ranks (WORLD_SIZE) / a process group of %d" % (args.gpus, world, dist.get_world_size()))
    backend = dist.get_backend()
    if not args.rehearse and backend != "nccl":
        raise SystemExit("bench.py --gpus N runs over RCCL (torch.distributed backend \"nccl\"); the process group's backend is %r" % backend)
    prep_s = 0.0
    lib_set = not args.strong and not args.replicas       # the default N > 1 job: N libraries, bin-sharded -- and the strong line beside it
    names = [name if r == 0 else "%s_s%d" % (name, 8 + r) for r in range(world)]
    if lib_set:
        # rank r prepares library r (seed 8 + r) while the others prepare theirs
        t0 = time.time()
        prepare_library(args.work, names[rank], args.reads, L, genome, 8 + rank, max(2, min(cores // world, 32)), args.paired)
        prep_s = time.time() - t0
    elif rank == 0:
        t0 = time.time()
        prepare_library(args.work, name, args.reads, L, genome, 8, min(cores, 32), args.paired)
        prep_s = time.time() - t0
    dist.barrier()
    binned_set = [os.path.join(args.work, n + ".b8") for n in names]
    binned = binned_set[0]
    fastq_one = int(open(os.path.join(args.work, name + ".done")).read())
    fastq_set = sum(int(open(os.path.join(args.work, n + ".done")).read()) for n in names) if lib_set else fastq_one

    sharded = not args.replicas
    threads = max(2, cores // world)
    packer = fastore_amd.Packer(device_id=local, lib=lib, host_threads=threads, rank=rank if sharded else 0, world_size=world if sharded else 1)
    coll_dev = None if args.rehearse else torch.device("cuda", local)
    keys = ["algorithmic_bytes", "ppmd_symbols", "host_coded_symbols", "kernel_launches", "encode_kernel_ms", "cdata_bytes", "bins", "records"]

    def drop(prefixes):
        for oo in prefixes:
            for e in (".cdata", ".cmeta"):
                try:
                    os.remove(oo + e)
                except OSError:
                    pass

    def timed(kind, steps, warmup):
        """W untimed + K timed steps of one job kind, bracketed by barrier + synchronize, MAX over the ranks"""
        made, removed = [], set()
        out_base = os.path.join(args.work, ("out_%s" % kind) if sharded else "out_r%d" % rank)

        def outs(o):
            return ["%s_l%d" % (o, i) for i in range(world)] if kind == "set" else [o]

        def step():
            if len(made) > 4 and (rank == 0 or not sharded):
                old = made[len(made) - 5]
                if old not in removed:
                    removed.add(old)
                    threading.Thread(target=drop, args=(outs(old),), daemon=True).start()
            o = out_base + "_%d" % len(made)
            made.append(o)
            if kind == "set":
                shard.pack_sharded_set(packer, binned_set, outs(o), dist, device=coll_dev)
            elif kind == "one":
                shard.pack_sharded(packer, binned, o, dist, device=coll_dev)
            else:
                packer.pack_file(binned, o)

        for _ in range(warmup):
            step()
        packer.reset_stats()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        st = packer.stats()
        t = torch.tensor([dt], device="cpu" if args.rehearse else "cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        v = torch.tensor([float(st[k]) for k in keys], device="cpu" if args.rehearse else "cuda", dtype=torch.float64); dist.all_reduce(v)
        if rank == 0 or not sharded:
            for o in made[:-1]:
                drop(outs(o))
        dist.barrier()
        return dt, dict(zip(keys, v.tolist())), st, outs(made[-1])

    # The headline of an N > 1 run is the ONE library of the N = 1 run bin-sharded over the ranks (BASELINE.json's configs[3] / [4]: strong
    # scaling; its step is bound by single streams, which more GPUs do not shorten -- reported as measured); the SET of N libraries
    # (weak scaling: per-GPU work fixed) is measured beside it and printed under `weak_set`.  --weak-set swaps the two.
    set_first = lib_set and args.weak_set
    main_kind = ("set" if set_first else "one") if sharded else "replica"
    dt, tot, st, last = timed(main_kind, args.steps, args.warmup)
    side = None; slast = None
    if lib_set:
        ks, kw = max(1, min(args.steps, 3)), 1
        side_kind = "one" if set_first else "set"
        sdt, stot, _, slast = timed(side_kind, ks, kw)
        sbytes = fastq_one if side_kind == "one" else fastq_set
        side = {"scaling": "strong" if side_kind == "one" else "weak", "value": round(sbytes * ks / sdt / 1e6, 2), "unit": "MB/s", "steps": ks, "warmup": kw, "ms_per_step": round(sdt / ks * 1e3, 2),
                "workload": ("ONE library (library 0 of the set, %.1f MB FASTQ) bin-sharded over the %d ranks" % (fastq_one / 1e6, world)) if side_kind == "one" else
                            ("a SET of %d libraries (%.1f MB FASTQ), every rank its share of every library's bins in one device pipeline" % (world, fastq_set / 1e6)),
                "roofline": roofline_of(stot, ks)}
    # (the archives of the two lines: `set_last` = the N archives of the set, `one_last` = the one library's)
    set_last = (last if set_first else slast) if lib_set else None
    one_last = (slast if set_first else last) if sharded else None

    # parity of the SET line: rank r runs the reference on library r (the ranks side by side, each with its share of the host's
    # cores) and compares the archive the N ranks wrote for it block for block; rank 0 collects the verdicts
    set_parity = None
    if lib_set and not args.no_cpu_baseline:
        pe_flag = ["-z"] if args.paired else []
        refp_r = os.path.join(args.work, "ref_" + names[rank])
        nt_r, tn_r, _ = reference_pack(binned_set[rank], refp_r, max(4, cores // world), pe_flag)
        ok_r = bool(nt_r is not None and same_blocks(set_last[rank], refp_r))
        strong_ok = bool(nt_r is not None and same_blocks(one_last[0], refp_r)) if rank == 0 else None
        for e in (".cdata", ".cmeta"):
            try:
                os.remove(refp_r + e)
            except OSError:
                pass
        gathered = [None] * world
        dist.all_gather_object(gathered, (ok_r, nt_r, tn_r, strong_ok))
        set_parity = gathered
    if rank == 0:
        jobs = world if args.replicas else 1
        fastq_bytes = fastq_set if main_kind == "set" else fastq_one
        value = fastq_bytes * jobs * args.steps / dt / 1e6
        rf = roofline_of(tot, args.steps); rf["ppmd_symbols_per_s_whole_job"] = round(tot["ppmd_symbols"] / dt, 1)
        sym = max(1.0, float(tot["ppmd_symbols"]))
        res = {
            "metric": "fastore_pack compressed MB/s (input FASTQ)", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak" if (main_kind == "set" or args.replicas) else "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s of %.1f M x %d bp %s synthetic FASTQ (gen_fastq genome %d bp, seed%s), --%s, C1 profile%s%s"
                                   % ("ONE library" if main_kind != "set" else "a SET of %d libraries, each" % world, args.reads / 1e6, L, "PE pairs" if args.paired else "SE reads", genome,
                                      " 8" if main_kind != "set" else "s 8..%d" % (7 + world), QUALITY,
                                      "" if not args.paired else " (configs[2] scaled by %g)" % (args.reads / 100e6),
                                      "" if main_kind != "one" else ": the library of the N = 1 line, bin-sharded over %d GPUs as configs[3] / configs[4] shard theirs" % world),
                       "fastq_bytes": fastq_bytes, "pack_flags": " ".join(PACK_FLAGS),
                       "parallelism": ("%d ranks, each the whole library (replicas)" % world if args.replicas else
                                       ("%d ranks pack disjoint LPT shards of the library's bins; one all-reduce of the block-size table over RCCL; no block bytes cross ranks" % world if main_kind == "one" else
                                        "%d ranks, each its LPT share of the bins of all %d libraries in one device pipeline; one all-reduce of the concatenated block-size tables over RCCL; no block bytes cross ranks" % (world, world)))},
            "roofline": rf,
            "host_coded_symbol_fraction": round(float(tot.get("host_coded_symbols", 0)) / sym, 4),
            "stages_ms_per_step_rank0": dict({k: round(st[k] / args.steps, 1) for k in ("encode_kernel_ms", "assemble_kernel_ms", "frontend_ms", "io_ms", "total_ms")}, block0_ms=round(st["block0_ms"], 1)),
            "archive": {"cdata_bytes": int(tot["cdata_bytes"]) // args.steps, "bins": int(tot["bins"]) // args.steps, "records": int(tot["records"]) // args.steps},
            "device": packer.device_name, "host_cores": cores, "prep_s": round(prep_s, 1),
            "ranks": world, "collective_backend": "%s (%s)" % (backend, "RCCL over xGMI" if backend == "nccl" else "CPU rehearsal on one device"),
        }
        if side is not None:
            res["strong" if side["scaling"] == "strong" else "weak_set"] = side
        if set_parity is not None:
            ok0, nt0, tn0, strong_ok = set_parity[0]
            if nt0 is not None:
                res["cpu_baseline"] = {"value": round(fastq_one / tn0 / 1e6, 2), "unit": "MB/s", "cores": min(nt0, max(4, cores // world)), "kind": "reference",
                                       "sample": "reference fastore_pack e -t%d on library 0 (%.1f MB FASTQ) while the other ranks run it on their libraries, %d host cores" % (nt0, fastq_one / 1e6, cores),
                                       "threads": nt0, "seconds": round(tn0, 2)}
            set_ok = {"every_library_every_block_bit_identical_to_reference": bool(all(g[0] for g in set_parity)), "per_library": [bool(g[0]) for g in set_parity],
                      "on": "all %d archives of the SET as the %d ranks wrote them, each against the reference's pack of that library" % (world, world)}
            one_ok = {"every_block_bit_identical_to_reference": bool(strong_ok), "block_order": "-t1 (block 0, ascending signature)",
                      "on": "the ONE library's archive as the %d ranks wrote it (%d blocks) against the reference's pack of it" % (world, len(read_archive(one_last[0])[0]))}
            # (the headline's parity on top, the other line's under its key)
            res["parity"] = dict(set_ok, one_library=one_ok) if main_kind == "set" else dict(one_ok, weak_set=set_ok)
        elif not args.no_cpu_baseline and sharded:
            # parity of the N > 1 job (--strong): the ONE library's archive against the reference's pack of it; the reference is
            # timed on rank 0's host cores while the other ranks wait
            pe = ["-z"] if args.paired else []
            refp = os.path.join(args.work, "ref_" + name)
            nt, tn, _ = reference_pack(binned, refp, cores, pe)
            if nt is not None:
                res["cpu_baseline"] = {"value": round(fastq_one / tn / 1e6, 2), "unit": "MB/s", "cores": min(nt, cores), "kind": "reference",
                                       "sample": "reference fastore_pack e -t%d on library 0 (%.1f MB FASTQ), %d host cores" % (nt, fastq_one / 1e6, cores), "threads": nt, "seconds": round(tn, 2)}
                res["parity"] = {"library_0_every_block_bit_identical_to_reference": bool(same_blocks(one_last[0], refp)),
                                 "on": "library 0 of the job as the %d ranks wrote it (%d blocks)" % (world, len(read_archive(one_last[0])[0]))}
            drop([refp])
        print(json.dumps(res), flush=True)
    packer.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
