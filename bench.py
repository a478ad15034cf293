#!/usr/bin/env python3
"""bench.py -- fastore_pack compressed MB/s (input FASTQ) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the pack hot path (`fastore_pack e`: read .b*, read-cluster modelling,
PPMd / range-coder entropy coding on the GPU, write .c*) over the workload below.  Prints ONE JSON line
(rank 0).  Workload (BASELINE.json configs[1]): 10 M x 150 bp single-end synthetic reads, --lossless,
C1 profile.  The reference's fastore_rebin does not scale with threads (~25 s per 1 M reads per pass on
any core count), so the 10 M reads are binned as TEN independent 1 M-read libraries prepared in parallel
by the real reference tools (oracle/_ref, untimed); all ten are packed as one job whose bins share the
device batches.  With --gpus N every rank packs the same prepared libraries into its own archives
(weak scaling: per-GPU work fixed; bins are independent, no data-path collective).
"""
import argparse
import concurrent.futures as cf
import json
import os
import subprocess
import sys
import time

# the pack context runs one HIP stream per pipeline slice; they need their own hardware queues to overlap
# (the runtime default is 4, shared round-robin) -- must be in the environment before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
REF_GCC = os.path.join(ROOT, "oracle", "_ref", "ref_driver_gcc")
GEN = os.path.join(ROOT, "build", "gen_fastq")
PACK_FLAGS = ["-r", "-f256", "-c10", "-d8", "-w1024", "-W1024"]
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy rate


def sh(cmd):
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def prepare_library(work, name, reads, length, genome, seed, threads, paired=False):
    """FASTQ -> fastore_bin -> 3 x fastore_rebin with the real reference (C1 profile). Cached."""
    base = os.path.join(work, name)
    binned = base + ".b8"
    pe = ["-z"] if paired else []
    if not (os.path.exists(binned + ".bmeta") and os.path.exists(base + ".done")):
        sh([GEN, "--reads", str(reads), "--len", str(length), "--genome", str(genome), "--seed", str(seed), "--out", base] + (["--paired"] if paired else []))
        inp = base + "_1.fastq" + ((" " + base + "_2.fastq") if paired else "")
        sh([REF_GCC, "bin", "-i" + inp, "-o" + base + ".b0", "-t%d" % threads, "-H", "-q0", "-p8", "-s0", "-b256"] + pe)
        prev = base + ".b0"
        for p in (2, 4, 8):
            cur = base + ".b%d" % p
            sh([REF_GCC, "rebin", "-i" + prev, "-o" + cur, "-t%d" % threads, "-r", "-w1024", "-W1024", "-p%d" % p] + pe)
            for e in ("bmeta", "bdna", "bqua", "bhead"):
                if os.path.exists(prev + "." + e):
                    os.remove(prev + "." + e)
            prev = cur
        open(base + ".done", "w").write("ok")
    return binned, os.path.getsize(base + "_1.fastq") + (os.path.getsize(base + "_2.fastq") if paired else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--libs", type=int, default=10, help="number of 1 M-read libraries (10 = configs[1])")
    ap.add_argument("--reads-per-lib", type=int, default=1_000_000)
    ap.add_argument("--work", default=os.environ.get("FASTORE_BENCH_DIR", "/tmp/fastore_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--paired", action="store_true", help="paired-end libraries (--reads-per-lib pairs each): configs[2]-shaped side measurement, not the default line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    os.makedirs(args.work, exist_ok=True)
    if not os.path.exists(GEN):
        subprocess.check_call(["g++", "-O2", "-o", GEN, os.path.join(ROOT, "tools", "gen_fastq.cpp")])
    have_ref = os.path.exists(REF) and os.path.exists(REF_GCC)
    if not have_ref:
        raise SystemExit("bench.py needs the reference tools under oracle/_ref (built by __graft_entry__.build()) to bin the synthetic FASTQ")

    L = 150
    cores = os.cpu_count() or 8
    libs = []
    if rank == 0:
        t0 = time.time()
        per = max(2, min(8, cores // max(1, args.libs + 1)))
        with cf.ThreadPoolExecutor(max_workers=max(1, min(args.libs + 1, cores // 2))) as ex:
            tag = "pe" if args.paired else "lib"
            cov = 2 if args.paired else 1        # bases per record: the genome is sized for ~50x coverage either way
            futs = [ex.submit(prepare_library, args.work, "%s%02d" % (tag, i), args.reads_per_lib, L, cov * args.reads_per_lib * L // 50, 8 + i, per, args.paired) for i in range(args.libs)]
            fs = ex.submit(prepare_library, args.work, "sample_pe" if args.paired else "sample", 200_000, L, cov * 200_000 * L // 50, 99, per, args.paired)
            libs = [f.result() for f in futs]
            sample = fs.result()
        prep_s = time.time() - t0
    if world > 1:
        dist.barrier()
        if rank != 0:
            tag = "pe" if args.paired else "lib"
            libs = [(os.path.join(args.work, "%s%02d.b8" % (tag, i)),
                     sum(os.path.getsize(os.path.join(args.work, "%s%02d_%d.fastq" % (tag, i, m))) for m in ((1, 2) if args.paired else (1,)))) for i in range(args.libs)]
    fastq_bytes = sum(s for _, s in libs)
    ins = [b for b, _ in libs]
    outs = [os.path.join(args.work, "out_r%d_%02d" % (rank, i)) for i in range(len(libs))]

    import fastore_amd
    # FASTORE_AMD_LIB: A/B runs of alternative builds of the library (kernel experiments); default = the in-tree build
    alt = os.environ.get("FASTORE_AMD_LIB")
    packer = fastore_amd.Packer(device_id=local if world > 1 else 0, lib=fastore_amd.load_library(alt) if alt else None)

    def step():
        packer.pack_files(ins, outs)

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
    if world > 1:
        t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    st = packer.stats()

    if rank == 0:
        value = fastq_bytes * world * args.steps / dt / 1e6
        launches = max(1, st["kernel_launches"])
        avg_launch_s = st["encode_kernel_ms"] / 1e3 / launches
        achieved = st["algorithmic_bytes"] / launches / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC passes cannot run inside this process; the committed summary of the
        # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command is reported per launch
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tf) and args.libs == 10 and args.reads_per_lib == 1_000_000 and not args.paired:
            traffic = json.load(open(tf))["hbm_bytes_per_step"] / (launches / args.steps)
        out = {
            "metric": "fastore_pack compressed MB/s (input FASTQ)", "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x (%d x %d bp %s synthetic FASTQ, --lossless, C1 profile) = %.1f M %s per GPU, packed as one job"
                                   % (len(libs), args.reads_per_lib, L, "PE" if args.paired else "SE", len(libs) * args.reads_per_lib / 1e6, "pairs" if args.paired else "reads"),
                       "fastq_bytes_per_gpu": fastq_bytes, "pack_flags": " ".join(PACK_FLAGS), "parallelism": "bins sharded per GPU; no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "fs_encode_streams", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic,
                         "traffic_unit": "bytes per launch (profiles/r01_hbm_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
                         "avg_launch_ms": round(avg_launch_s * 1e3, 3), "launches": launches,
                         "algorithmic_bytes_per_launch": st["algorithmic_bytes"] // launches,
                         # the launches of a step overlap (one per pipeline slice, each on its own HIP stream), so a launch's
                         # duration includes the time it shares the GPU; the whole-GPU symbol rate is quoted per step wall time
                         "overlapping_launches_per_step": launches // args.steps,
                         "ppmd_symbols_per_s_whole_gpu": round(st["ppmd_symbols"] / dt, 1)},
            "stages_ms_per_step": {k: round(st[k] / args.steps, 1) for k in ("encode_kernel_ms", "assemble_kernel_ms", "frontend_ms", "io_ms", "block0_ms", "total_ms")},
            "archive": {"cdata_bytes": st["cdata_bytes"] // args.steps, "bins": st["bins"] // args.steps, "records": st["records"] // args.steps,
                        "block0_records": st["block0_records"] // args.steps},
            "device": packer.device_name, "host_cores": cores, "prep_s": round(prep_s, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            # reference CPU fastore_pack on a bounded sample (200 k x 150 bp, same generator/profile), and parity on it
            sb, sbytes = sample
            sp = os.path.join(args.work, "sample_pe" if args.paired else "sample")
            pe = ["-z"] if args.paired else []
            t = time.perf_counter(); sh([REF, "pack", "-i" + sb, "-o" + sp + ".t1", "-t1"] + PACK_FLAGS + pe); t1 = time.perf_counter() - t
            # the reference's multi-threaded pack dead-locks at -t64 (observed here and in the build container), so the
            # all-cores leg uses at most 32 workers, under a timeout, stepping down if it still hangs
            nt, tn = None, None
            for cand in (32, 16, 8):
                if cand > cores:
                    continue
                try:
                    t = time.perf_counter()
                    subprocess.run([REF, "pack", "-i" + sb, "-o" + sp + ".tn", "-t%d" % cand] + PACK_FLAGS + pe, stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, timeout=180, check=True)
                    nt, tn = cand, time.perf_counter() - t
                    break
                except (subprocess.TimeoutExpired, subprocess.CalledProcessError):
                    continue
            if nt is None:
                nt, tn = 1, t1
            packer.pack_file(sb, sp + ".gpu")
            same = open(sp + ".gpu.cdata", "rb").read() == open(sp + ".t1.cdata", "rb").read()
            out["cpu_baseline"] = {"value": round(sbytes / tn / 1e6, 2), "unit": "MB/s", "cores": nt, "kind": "reference",
                                   "sample": "reference fastore_pack e -t%d on 200 k x 150 bp %s of the same generator (%.1f MB FASTQ)" % (nt, "pairs" if args.paired else "SE", sbytes / 1e6),
                                   "t1_value": round(sbytes / t1 / 1e6, 2), "t1_seconds": round(t1, 2), "tn_seconds": round(tn, 2)}
            out["parity"] = {"cdata_bit_identical_to_reference_t1": bool(same), "on": "the cpu_baseline sample"}
        print(json.dumps(out), flush=True)
    packer.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
