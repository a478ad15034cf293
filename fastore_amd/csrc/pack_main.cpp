// fastore_pack -- command-line front of the MI355X-native pack path.
// Keeps the reference's `fastore_pack e` surface (fastore_pack/main.cpp:26-41, 165-330): first
// argument e|d, >= 3 arguments, -i<prefix> -o<prefix> -t<n> [-z] [-v] and the matcher/consensus
// knobs; errors as "Error: <what>" on stderr with exit status 255.  `d` (decode) is outside this
// build's scope: the reference binary named by FASTORE_PACK_REF is started as a child process (never exec'ed over this
// one: a process that may have initialised the GPU -- e.g. under a preloaded profiler -- must not be replaced).
// New flags: -g<device> (HIP device ordinal), -j<n> (host threads beyond the reference's -t limit of 64),
// -R<rank>/-N<world> (bin sharding: this process packs its share into <out>.part<rank>), -G<n> (one process, n GPUs:
// devices g .. g+n-1 each pack their LPT share of the bins side by side and write their blocks into the one archive).
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <spawn.h>
#include <sys/wait.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "../../include/fastore_amd.h"

static void usage()
{
    fprintf(stderr, "\n\n\t\t--- FaStore (MI355X pack path) ---\n\n\n"
                    "fastore_pack -- FASTQ reads compression tool\n\n"
                    "usage:\tfastore_pack <e|d> [options] -i<input_file> -o<output_file>\n"
                    "\t-i<file>\t: input file(s) prefix\t-o<file>\t: output files prefix\n"
                    "\t-z\t\t: use paired-end mode\n"
                    "\t-f<n> -e<n> -m<n> -s<n> -w<n> -r -l -E<n> -W<n> -c<n> -q<n> -n<n> -d<n>\t: as the reference\n"
                    "\t-t<n>\t\t: host threads\n\t-g<n>\t\t: HIP device ordinal\t-G<n>\t\t: number of devices (from -g on)\n\t-v\t\t: verbose mode\n");
}

int main(int argc, char** argv)
{
    if (argc < 1 + 3 || (argv[1][0] != 'e' && argv[1][0] != 'd')) { usage(); return 255; }
    if (argv[1][0] == 'd') {
        const char* ref = getenv("FASTORE_PACK_REF");
        if (!ref) { fprintf(stderr, "Error: decode mode is served by the reference fastore_pack; set FASTORE_PACK_REF to its path\n"); return 255; }
        extern char** environ;
        argv[0] = (char*)ref;
        pid_t pid = 0;
        const int rc = posix_spawn(&pid, ref, nullptr, nullptr, argv, environ);
        if (rc != 0) { fprintf(stderr, "Error: cannot start %s: %s\n", ref, strerror(rc)); return 255; }
        int status = 0;
        while (waitpid(pid, &status, 0) < 0) { if (errno != EINTR) { perror("Error: waitpid"); return 255; } }
        return WIFEXITED(status) ? WEXITSTATUS(status) : 255;
    }
    fsgpu_config cfg; fsgpu_config_defaults(&cfg);
    cfg.one_shot = 1;
    std::string in, out; int verbose = 0; int threads = 0, hostThreads = -1, gpus = 1; bool pe = false, threadsGiven = false;
    for (int i = 2; i < argc; ++i) {
        const char* p = argv[i];
        if (p[0] != '-') continue;
        const size_t len = strlen(p);
        int v = -1;
        if (len > 2 && len < 10) v = atoi(p + 2);
        switch (p[1]) {
        case 'i': in = p + 2; break;
        case 'o': out = p + 2; { size_t sp = out.find_first_of(" \n"); if (sp != std::string::npos) out = out.substr(0, sp); } break;
        case 't': threads = v; threadsGiven = true; break;
        case 'v': verbose = 1; break;
        case 'z': pe = true; break;
        case 'f': cfg.min_bin_size = v; break;
        case 'w': cfg.max_lz_window = v; break;
        case 'W': cfg.max_pair_lz_window = v; break;
        case 'e': cfg.encode_threshold = v; break;
        case 'E': cfg.pair_encode_threshold = v; break;
        case 's': cfg.shift_cost = v; break;
        case 'm': cfg.mismatch_cost = v; break;
        case 'r': cfg.extra_reduce_hard_reads = 1; break;
        case 'l': cfg.extra_reduce_expensive_lz = 1; break;
        case 'q': cfg.max_record_shift_diff = v; break;
        case 'n': cfg.max_new_variants_per_read = v; break;
        case 'd': cfg.max_hamming_distance = v; break;
        case 'c': cfg.min_consensus_size = v; break;
        case 'g': cfg.device_id = v; break;
        case 'G': gpus = v; break;
        case 'j': hostThreads = v; break;
        case 'R': cfg.rank = v; break;
        case 'N': cfg.world_size = v; break;
        // accepted by the reference's parser, without effect on `e` with a binned input (QVZ training options of the bin
        // stage, dry-run / FASTQ output of the decoder): said so instead of silently dropped
        case 'U': case 'F': case 'M': case 'T': case 'D':
            fprintf(stderr, "Warning: option -%c has no effect on the pack step and is ignored\n", p[1]); break;
        }
    }
    (void)pe;   // the read type is taken from the .bmeta config, as the reference effectively does for the data path
    if (in.empty()) { fprintf(stderr, "Error: no input file specified\n"); return 255; }
    if (out.empty()) { fprintf(stderr, "Error: no output file(s) specified\n"); return 255; }
    // fastore_pack/main.cpp:326: 1 <= t <= 64 (no -t at all = this build's default, all cores)
    if (threads < 0 || threads > 64 || (threadsGiven && threads == 0)) { fprintf(stderr, "Error: invalid number of threads specified\n"); return 255; }
    cfg.host_threads = hostThreads >= 0 ? (uint32_t)hostThreads : (uint32_t)threads;   // -j overrides -t (which keeps the reference's 1..64 range)
    if (gpus < 1 || gpus > 64) { fprintf(stderr, "Error: invalid number of devices specified\n"); return 255; }
    if (gpus > 1) {
        // one context (and one set of host threads) per device; every device takes its LPT share of the bins (longest first over
        // the .bmeta record totals, packer.cpp: shardOwners), no block bytes cross devices
        std::vector<fsgpu_ctx*> ctxs(gpus, nullptr);
        for (int r = 0; r < gpus; ++r) {
            fsgpu_config c = cfg; c.device_id = cfg.device_id + r; c.rank = (uint32_t)r; c.world_size = (uint32_t)gpus;
            // the host cores are shared by the contexts: each gets its share of -t/-j, or of the machine
            const uint32_t budget = c.host_threads ? c.host_threads : std::max(1u, std::thread::hardware_concurrency());
            c.host_threads = std::max(1u, (budget + (uint32_t)gpus - 1) / (uint32_t)gpus);
            ctxs[r] = fsgpu_create(&c);
            if (!ctxs[r]) { fprintf(stderr, "Error: %s\n", fsgpu_create_error()); for (fsgpu_ctx* x : ctxs) if (x) fsgpu_destroy(x); return 255; }
        }
        // every context codes its LPT share of the bins and holds the blocks; the size tables are summed here (between
        // processes that sum is the one collective, fastore_amd/shard.py); then every context writes its blocks at their
        // places in the one archive.  No part files, no block bytes between devices.
        std::vector<int> rcs(gpus, 0); std::vector<std::thread> th; std::vector<size_t> nb(gpus, 0);
        for (int r = 0; r < gpus; ++r) th.emplace_back([&, r]() { rcs[r] = fsgpu_shard_pack(ctxs[r], in.c_str(), &nb[r]); });
        for (auto& t : th) t.join();
        th.clear();
        int bad = -1; for (int r = 0; r < gpus; ++r) if (rcs[r] != 0 && bad < 0) bad = r;
        std::vector<uint64_t> all;
        if (bad < 0) {
            all.assign(nb[0], 0);
            std::vector<uint32_t> sg(nb[0]); std::vector<uint64_t> sz(nb[0]);
            for (int r = 0; r < gpus && bad < 0; ++r) {
                if (nb[r] != nb[0] || fsgpu_shard_table(ctxs[r], sg.data(), sz.data(), nb[0]) != 0) { bad = r; break; }
                for (size_t i = 0; i < nb[0]; ++i) all[i] += sz[i];
            }
        }
        if (bad < 0) {
            for (int r = 0; r < gpus; ++r) th.emplace_back([&, r]() { rcs[r] = fsgpu_shard_write(ctxs[r], out.c_str(), all.data(), all.size()); });
            for (auto& t : th) t.join();
            for (int r = 0; r < gpus; ++r) if (rcs[r] != 0 && bad < 0) bad = r;
        }
        if (bad >= 0) fprintf(stderr, "Error: %s\n", fsgpu_last_error(ctxs[bad]));
        for (fsgpu_ctx* x : ctxs) fsgpu_destroy(x);
        if (bad >= 0) { remove((out + ".cdata").c_str()); remove((out + ".cmeta").c_str()); return 255; }
        if (verbose) {
            char err[256] = {0};
            fprintf(stderr, "\rParts processed: %zu (100%%) \n", all.size());
            if (fsgpu_print_stream_sizes(out.c_str(), err, sizeof err) != 0) { fprintf(stderr, "Error: %s\n", err); return 255; }
        }
        return 0;
    }
    // A library of millions of reads has bins of tens of thousands: its pack is bound by single long streams, which a
    // thousand resident coder waves serve as well as three thousand (profiles/r02_oo_max_waves_sweep.txt) -- while every
    // gigabyte of arena costs a process that packs once set-up time (53 GB: 0.2 s more, profiles/r02_aj_back_to_back.txt).
    // The record count is in the first 40 bytes of .bmeta (fastore_bin/BinFile.h:106-118).
    if (cfg.max_waves == 0) {
        if (FILE* f = fopen((in + ".bmeta").c_str(), "rb")) {
            uint64_t h[2] = {0, 0};
            if (fread(h, 8, 2, f) == 2 && h[1] >= 3000000ull) cfg.max_waves = 1024;
            fclose(f);
        }
    }
    const bool trace = getenv("FS_TRACE") != nullptr;
    auto clk = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; };
    const double tm0 = clk();
    auto wallNow = []() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000; };
    if (trace) fprintf(stderr, "[trace] main: entered at %lld ms (epoch)\n", wallNow());
    fsgpu_ctx* ctx = fsgpu_create(&cfg);
    if (!ctx) { fprintf(stderr, "Error: %s\n", fsgpu_create_error()); return 255; }
    const double tm1 = clk();
    const int rc = fsgpu_pack_file(ctx, in.c_str(), out.c_str(), verbose);
    const double tm2 = clk();
    if (rc != 0) {
        fprintf(stderr, "Error: %s\n", fsgpu_last_error(ctx)); fsgpu_destroy(ctx);
        remove((out + ".cdata").c_str()); remove((out + ".cmeta").c_str());       // (no half-written archive is left behind)
        return 255;
    }
    if (verbose) {
        fsgpu_stats st; fsgpu_get_stats(ctx, &st);
        fprintf(stderr, "device %s: %llu bins, %llu records, encode kernel %.1f ms, assemble %.1f ms, front end %.1f ms, block0 %.1f ms, io %.1f ms, total %.1f ms\n",
                fsgpu_device_name(ctx), (unsigned long long)st.bins, (unsigned long long)st.records, st.encode_kernel_ms, st.assemble_kernel_ms,
                st.frontend_ms, st.block0_ms, st.io_ms, st.total_ms);
    }
    // The archive is on disk and closed.  A one-shot process does not hand its device and pinned memory back piece by
    // piece (0.75 s for a 10 M-read library: 53 GB of arenas, the lanes' pinned staging buffers): the kernel reclaims
    // them with the process.  FS_ORDERLY_EXIT=1 keeps the orderly teardown (leak checkers, the sanitizer builds).
    // A preloaded tool (rocprofv3, a coverage or leak checker) writes its output from exit handlers: take the orderly way then.
    if (!getenv("FS_ORDERLY_EXIT") && !getenv("LD_PRELOAD") && !getenv("ROCP_TOOL_LIBRARIES") && !getenv("ROCPROFILER_REGISTER_FORCE_LOAD")) {
        if (trace) fprintf(stderr, "[trace] main: context %.0f ms (HIP start-up, arena pool), pack %.0f ms, no teardown; leaving at %lld ms (epoch)\n", tm1 - tm0, tm2 - tm1, wallNow());
        fflush(stdout); fflush(stderr);
        _exit(0);
    }
    fsgpu_destroy(ctx);
    if (trace) fprintf(stderr, "[trace] main: context %.0f ms (HIP start-up, arena pool), pack %.0f ms, teardown %.0f ms\n", tm1 - tm0, tm2 - tm1, clk() - tm2);
    return 0;
}
