// extern "C" boundary (include/fastore_amd.h).  No exceptions cross it.
#include <stdlib.h>
#include <sched.h>
#include <string.h>
#include <unistd.h>
#include <stdexcept>
#include <string>
#include <future>
#include <thread>
#include "../../include/fastore_amd.h"
#include "packer.h"

struct fsgpu_ctx { fs::Context c; std::vector<fsgpu_ctx*> helpers; };      // helpers: the further pipelines of a library of several batches (packSplit)

static thread_local std::string g_createError;

#define FS_GUARD(ctx, ...)                                                               \
    try { __VA_ARGS__; return FSGPU_OK; }                                                       \
    catch (const std::exception& e) { (ctx)->c.err = e.what();                          \
        return strncmp(e.what(), "device:", 7) == 0 ? FSGPU_ERR_DEVICE : (strncmp(e.what(), "Cannot", 6) == 0 ? FSGPU_ERR_IO : FSGPU_ERR_INTERNAL); } \
    catch (...) { (ctx)->c.err = "unknown error"; return FSGPU_ERR_INTERNAL; }

extern "C" {

void fsgpu_config_defaults(fsgpu_config* cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    fs::PackParams p;
    cfg->min_bin_size = p.minBinSize; cfg->shift_cost = p.shiftCost; cfg->mismatch_cost = p.mismatchCost;
    cfg->max_lz_window = p.maxLzWindowSize; cfg->max_pair_lz_window = 4096;    // MAX_LZ_PE (fastore_bin/Globals.h:62)
    cfg->max_new_variants_per_read = p.maxNewVariantsPerRead; cfg->max_hamming_distance = p.maxHammingDistance;
    cfg->min_consensus_size = p.minConsensusSize; cfg->world_size = 1;
}

int fsgpu_device_count(void) { return fsengine::device_count(); }
const char* fsgpu_create_error(void) { return g_createError.c_str(); }

// packSplit -> fsgpu_create: the device whose arena pool the context being made shares (one pool a device for the pipelines of a split pack: three
// pools were 53 GB that the next process's allocations waited for the driver to take back -- 2.9 s of a 25 M-pair pack behind another)
static thread_local fsengine::Device* t_poolDonor = nullptr;

static unsigned usableCores()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}

fsgpu_ctx* fsgpu_create(const fsgpu_config* cfg)
{
    if (!cfg) { g_createError = "null config"; return nullptr; }
    fsgpu_ctx* ctx = new fsgpu_ctx();
    fs::Context& c = ctx->c;
    c.cfg = *cfg;
    c.par.minBinSize = cfg->min_bin_size; c.par.encodeThreshold = cfg->encode_threshold; c.par.pairEncodeThreshold = cfg->pair_encode_threshold;
    c.par.shiftCost = cfg->shift_cost; c.par.mismatchCost = cfg->mismatch_cost;
    c.par.maxLzWindowSize = cfg->max_lz_window; c.par.maxPairLzWindowSize = cfg->max_pair_lz_window;
    c.par.extraReduceHardReads = cfg->extra_reduce_hard_reads != 0; c.par.extraReduceExpensiveLzMatches = cfg->extra_reduce_expensive_lz != 0;
    c.par.maxRecordShiftDifference = cfg->max_record_shift_diff; c.par.maxNewVariantsPerRead = cfg->max_new_variants_per_read;
    c.par.maxHammingDistance = cfg->max_hamming_distance; c.par.minConsensusSize = cfg->min_consensus_size;
    // default: 1.5 host threads per core this process may really use (the smaller of the hardware count, the affinity
    // mask and the cgroup CPU quota), at most 24.  The measurement box shows 256 CPUs but its container is capped at 16
    // cores: measured 12 / 14 / 16 / 20 / 24 / 32 / 48 threads -> 24 is best (the workers compete with the lane, block-0
    // and runtime threads for the same 16 cores), more only time-slice.
    // Ranks of one node (one process per GPU under torch.distributed.run, which exports LOCAL_WORLD_SIZE) share the cores.
    unsigned localRanks = 1;
    if (const char* lw = getenv("LOCAL_WORLD_SIZE")) localRanks = (unsigned)std::max(1, atoi(lw));
    const unsigned share = usableCores() * 3u / 2u / localRanks;
    c.hostCores = std::max(1u, usableCores() / localRanks);
    c.hostThreads = cfg->host_threads ? cfg->host_threads : std::max(localRanks > 1 ? 4u : 1u, std::min(24u, share));
    if (getenv("FS_BATCH_BASES") && atoll(getenv("FS_BATCH_BASES")) > 0) c.cfg.batch_bases = (uint64_t)atoll(getenv("FS_BATCH_BASES"));      // (tests: small device batches)
    if (getenv("FS_PIPELINE_SLICES") && atoi(getenv("FS_PIPELINE_SLICES")) > 0) c.cfg.pipeline_slices = (uint32_t)atoi(getenv("FS_PIPELINE_SLICES"));
    if (getenv("FS_PIPELINE_LANES") && atoi(getenv("FS_PIPELINE_LANES")) > 0) c.cfg.pipeline_lanes = (uint32_t)atoi(getenv("FS_PIPELINE_LANES"));
    if (getenv("FS_DEVICE_MATCHER") && atoi(getenv("FS_DEVICE_MATCHER")) == 0) c.deviceMatcher = false;       // A/B runs: the host window scan
    if (getenv("FS_MAX_WAVES") && atoi(getenv("FS_MAX_WAVES")) > 0) c.cfg.max_waves = (uint32_t)atoi(getenv("FS_MAX_WAVES"));
    if (c.par.mismatchCost <= 0 || c.par.shiftCost < 0 || c.par.maxLzWindowSize == 0 || c.par.maxPairLzWindowSize == 0) { g_createError = "invalid matcher parameters"; delete ctx; return nullptr; }
    fsengine::set_pageable_staging(cfg->one_shot != 0);       // one-shot contexts stage through pageable memory
    // the device: at once -- or, for a context that packs once, on a thread of its own while the caller goes on to the
    // pack call and the front end of the first bins (fs::Context::device() waits for it; a failure is reported there, as
    // loudly: "no HIP device available ..." becomes the pack call's error).
    fs::Context* cp = &c;
    const int deviceId = cfg->device_id; const uint32_t maxWaves = c.cfg.max_waves;
    fsengine::Device* const donor = t_poolDonor;      // (a helper pipeline of a split pack: its lanes are made on the parent's arena pool)
    auto make = [cp, deviceId, maxWaves, donor]() -> std::string {
        if (fsengine::device_count() <= 0) return "no HIP device available: the fastore_pack hot path has no CPU fallback";
        char err[256] = {0};
        if (donor) { if (fsengine::lane_create(donor, &cp->dev, err, sizeof err) != 0) return err[0] ? std::string(err) : std::string("lane creation failed"); return std::string(); }
        if (fsengine::device_create(&cp->dev, deviceId, maxWaves, err, sizeof err) != 0) return err[0] ? std::string(err) : std::string("device creation failed");
        return std::string();
    };
    if (cfg->one_shot) {
        c.devPending = std::async(std::launch::async, make).share();
        c.devAsync.store(true);
        return ctx;
    }
    const std::string e = make();
    if (!e.empty()) { g_createError = e; delete ctx; return nullptr; }
    return ctx;
}

void fsgpu_destroy(fsgpu_ctx* ctx)
{
    if (!ctx) return;
    if (ctx->c.devAsync.load()) { try { (void)ctx->c.device(); } catch (...) {} }       // (a device still on its way is waited for, then released)
    const bool trace = getenv("FS_TRACE") != nullptr;
    auto clk = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; };
    const double t0 = clk();
    // (the pipelines a large library was split over hold tens of gigabytes of host pages each: they are given back side by side, and beside
    // this context's own -- one after the other it was 1.2-1.5 s behind a 25 M-pair pack, profiles/r05_cli_trace_pe_25m.txt)
    std::vector<std::thread> helperEnds;
    for (fsgpu_ctx* h : ctx->helpers) helperEnds.emplace_back([h]() { fsgpu_destroy(h); });
    ctx->helpers.clear();
    struct JoinAll { std::vector<std::thread>& t; ~JoinAll() { for (auto& x : t) if (x.joinable()) x.join(); } } joinHelpers{helperEnds};
    for (fsengine::MatchLane* m : ctx->c.matchLanes) fsengine::match_lane_destroy(m);
    const double t1 = clk();
    for (size_t i = 1; i < ctx->c.lanes.size(); ++i) fsengine::device_destroy(ctx->c.lanes[i]);      // lanes[0] == dev
    const double t2 = clk();
    fsengine::device_destroy(ctx->c.dev);
    const double t3 = clk();
    delete ctx;
    if (trace) fprintf(stderr, "[trace] destroy: matcher lanes %.1f ms, lanes %.1f ms, first lane + pool %.1f ms, host buffers %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, clk() - t3);
}
const char* fsgpu_last_error(const fsgpu_ctx* ctx) { return ctx ? ctx->c.err.c_str() : "null context"; }
const char* fsgpu_device_name(const fsgpu_ctx* ctx)
{
    if (!ctx) return "";
    try { return const_cast<fsgpu_ctx*>(ctx)->c.device()->name; } catch (...) { return ""; }
}

int fsgpu_set_archive_params(fsgpu_ctx* ctx, const void* cfgRaw, size_t cfgBytes, const uint8_t* fields, size_t fieldBytes)
{
    if (!ctx || !cfgRaw || cfgBytes != sizeof(fs::BinModuleConfigRaw)) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        ctx->c.archives.assign(1, fs::ArchiveParams());
        fs::ArchiveParams& a = ctx->c.archives[0];
        memcpy(&a.cfg, cfgRaw, cfgBytes);
        fs::parseHeaderFields(fields, fieldBytes, a.cfg.archiveType.readType == fs::READ_PE, a.head);
        if (a.cfg.archiveType.readsHaveHeaders && a.head.fields.empty()) throw std::runtime_error("archive has read ids but no field table was given");
        ctx->c.haveArchive = true;
    });
}

int fsgpu_set_quality_codebook(fsgpu_ctx* ctx, const uint8_t* footer, size_t bytes)
{
    if (!ctx || !footer) return FSGPU_ERR_ARG;
    if (!ctx->c.haveArchive) { ctx->c.err = "fsgpu_set_archive_params() has not been called"; return FSGPU_ERR_ARG; }
    FS_GUARD(ctx, {
        fs::BitReader r(footer, bytes);
        ctx->c.archives[0].qvz = fs::QvzModel();
        ctx->c.archives[0].qvz.parse(r);
    });
}

int fsgpu_qvz_encode(fsgpu_ctx* ctx, const uint8_t* footer, size_t footerBytes, size_t n, const uint8_t* const* quals,
                     const uint32_t* const* readLens, const size_t* nReads, uint8_t* const* out, const size_t* outCap, size_t* outLen)
{
    using namespace fsdev;
    if (!ctx || !footer || (n && (!quals || !readLens || !nReads || !out || !outCap || !outLen))) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        fs::BitReader r(footer, footerBytes);
        fs::QvzModel model; model.parse(r);
        std::vector<StreamItem> items(n);
        std::vector<std::vector<uint8_t>> sym(n);
        uint64_t inBytes = (model.blob.size() + 15) & ~15ull;
        for (size_t i = 0; i < n; ++i) {
            fs::WellRng rng; rng.reset(model.wellSeed);
            const uint8_t* q = quals[i];
            for (size_t k = 0; k < nReads[i]; ++k) { fs::qvzSymbolise(model, rng, q, readLens[i][k], 0, false, sym[i]); q += readLens[i][k]; }
            if (sym[i].size() > 0xF0000000ull) throw std::runtime_error("stream larger than 4 GiB");
            StreamItem it; memset(&it, 0, sizeof it);
            it.kind = KIND_QVZ; it.in_len = (uint32_t)(sym[i].size() / 4); it.in_off = inBytes; it.aux_off = 0; it.out_cap = 3 * it.in_len + 64;
            items[i] = it; inBytes += (sym[i].size() + 15) & ~15ull;
        }
        std::vector<uint8_t> input(inBytes + 16);
        memcpy(input.data(), model.blob.data(), model.blob.size());
        for (size_t i = 0; i < n; ++i) if (!sym[i].empty()) memcpy(input.data() + items[i].in_off, sym[i].data(), sym[i].size());
        std::vector<uint8_t> raw; std::vector<uint32_t> rawSizes;
        if (fsengine::encode_streams_raw(ctx->c.device(), input.data(), inBytes, items, raw, rawSizes, &ctx->c.timing) != 0) throw std::runtime_error(std::string("device: ") + ctx->c.dev->err);
        for (size_t i = 0; i < n; ++i) {
            if (rawSizes[i] == 0xFFFFFFFFu) throw std::runtime_error("QVZ stream: malformed symbol or output overflow");
            outLen[i] = rawSizes[i];
            const size_t c = std::min<size_t>(rawSizes[i], outCap[i]);
            if (c) memcpy(out[i], raw.data() + items[i].out_off, c);
        }
    });
}

int fsgpu_compress_bins(fsgpu_ctx* ctx, const fsgpu_bin_batch* in, fsgpu_block_batch* out)
{
    if (!ctx || !in || !out) return FSGPU_ERR_ARG;
    if (!ctx->c.haveArchive) { ctx->c.err = "fsgpu_set_archive_params() has not been called"; return FSGPU_ERR_ARG; }
    FS_GUARD(ctx, {
        fs::Batch b;
        const uint32_t sigLen = ctx->c.archives[0].cfg.minimizer.signatureLen;
        if ((in->n_bases && (!in->bases || !in->quals)) || (in->n_heads && !in->heads) || (in->n_records && !in->records) || (in->n_nodes && !in->nodes)
            || (in->n_top_nodes && !in->top_nodes) || (in->n_em_records && !in->em_records) || (in->n_trees && !in->trees) || (in->n_bins && !in->bins))
            throw std::runtime_error("null array in the bin batch");
        b.seq.assign(in->bases, in->bases + in->n_bases); b.qua.assign(in->quals, in->quals + in->n_bases);
        if (in->n_heads) b.head.assign(in->heads, in->heads + in->n_heads);
        b.recs.resize(in->n_records);
        for (size_t i = 0; i < in->n_records; ++i) {
            const fsgpu_record& r = in->records[i];
            if ((uint64_t)r.seq_off + r.seq_len + r.aux_len > in->n_bases || (uint64_t)r.head_off + r.head_len > in->n_heads) throw std::runtime_error("record outside the batch buffers");
            if ((uint32_t)r.minim_pos + sigLen > r.seq_len) throw std::runtime_error("record with its signature outside the read");
            b.recs[i] = fs::Rec{r.seq_off, r.head_off, r.seq_len, r.aux_len, r.minim_pos, r.head_len, r.flags};
        }
        b.nodes.resize(in->n_nodes);
        for (size_t i = 0; i < in->n_nodes; ++i) {
            const fsgpu_node& n = in->nodes[i];
            if (n.rec >= in->n_records || (uint64_t)n.em_begin + n.em_count > in->n_em_records || (uint64_t)n.tree_begin + n.tree_count > in->n_trees) throw std::runtime_error("node references outside the batch");
            // sub-tree nodes come after their owner (NodesPacker.cpp:567-679 writes them depth first): anything else could
            // make the graph cyclic
            for (uint32_t t = 0; t < n.tree_count; ++t) if (in->trees[n.tree_begin + t].node_begin <= i) throw std::runtime_error("sub-tree nodes must follow their owner");
            b.nodes[i] = fs::NodeIn{n.rec, n.em_begin, n.em_count, n.tree_begin, n.tree_count};
        }
        b.topNodes.assign(in->top_nodes, in->top_nodes + in->n_top_nodes);
        for (uint32_t t : b.topNodes) if (t >= in->n_nodes) throw std::runtime_error("top node outside the batch");
        b.emRecs.assign(in->em_records, in->em_records + in->n_em_records);
        for (uint32_t e : b.emRecs) if (e >= in->n_records) throw std::runtime_error("exact-match record outside the batch");
        b.trees.resize(in->n_trees);
        for (size_t i = 0; i < in->n_trees; ++i) {
            const fsgpu_tree& t = in->trees[i];
            if ((uint64_t)t.node_begin + t.node_count > in->n_nodes) throw std::runtime_error("tree nodes outside the batch");
            b.trees[i] = fs::TreeIn{t.signature, t.main_signature_pos, t.node_begin, t.node_count};
        }
        b.bins.resize(in->n_bins);
        for (size_t i = 0; i < in->n_bins; ++i) {
            const fsgpu_bin& x = in->bins[i];
            if ((uint64_t)x.rec_begin + x.rec_count > in->n_records || (uint64_t)x.top_begin + x.top_count > in->n_top_nodes || x.top_count == 0) throw std::runtime_error("bin ranges outside the batch");
            fs::BinIn bi{}; bi.signature = x.signature; bi.minLen = x.min_len; bi.maxLen = x.max_len; bi.rawDnaSize = x.raw_dna_size;
            bi.recBegin = x.rec_begin; bi.recCount = x.rec_count; bi.topBegin = x.top_begin; bi.topCount = x.top_count;
            b.bins[i] = bi;
        }
        ctx->c.compressBatch(b, std::vector<uint32_t>(b.bins.size(), 0u));
        ctx->c.gatherBlocks();
        out->data = ctx->c.blocks.data(); out->sizes = ctx->c.blockSizes.data(); out->n_blocks = ctx->c.blockSizes.size();
    });
}

struct fsgpu_library {
    fs::Batch batch;
    fs::BinModuleConfigRaw cfg;
    std::vector<uint8_t> fields, qvz;
    std::vector<fsgpu_record> records; std::vector<fsgpu_node> nodes; std::vector<fsgpu_tree> trees; std::vector<fsgpu_bin> bins;
    fsgpu_bin_batch view;
};

fsgpu_library* fsgpu_library_open(const char* inPrefix, uint32_t minBinSize)
{
    if (!inPrefix) { g_createError = "null prefix"; return nullptr; }
    fsgpu_library* lib = nullptr;
    try {
        lib = new fsgpu_library();
        fs::BinFile bf; bf.open(inPrefix, minBinSize);
        lib->cfg = bf.config();
        if (lib->cfg.archiveType.readsHaveHeaders) fs::serializeHeaderFields(bf.headData(), lib->cfg.archiveType.readType == fs::READ_PE, lib->fields);
        if (bf.qvz().present) lib->qvz = bf.qvz().footerBytes;
        fs::Batch& b = lib->batch;
        for (uint32_t sig : bf.stdSignatures()) bf.unpack(sig, b, true);
        lib->records.resize(b.recs.size());
        for (size_t i = 0; i < b.recs.size(); ++i) { const fs::Rec& r = b.recs[i]; lib->records[i] = fsgpu_record{r.seqOff, r.headOff, r.seqLen, r.auxLen, r.minimPos, r.headLen, r.flags}; }
        lib->nodes.resize(b.nodes.size());
        for (size_t i = 0; i < b.nodes.size(); ++i) { const fs::NodeIn& n = b.nodes[i]; lib->nodes[i] = fsgpu_node{n.rec, n.emBegin, n.emCount, n.treeBegin, n.treeCount}; }
        lib->trees.resize(b.trees.size());
        for (size_t i = 0; i < b.trees.size(); ++i) { const fs::TreeIn& t = b.trees[i]; lib->trees[i] = fsgpu_tree{t.signatureId, t.mainSignaturePos, t.nodeBegin, t.nodeCount}; }
        lib->bins.resize(b.bins.size());
        for (size_t i = 0; i < b.bins.size(); ++i) { const fs::BinIn& x = b.bins[i]; lib->bins[i] = fsgpu_bin{x.signature, x.minLen, x.maxLen, x.rawDnaSize, x.recBegin, x.recCount, x.topBegin, x.topCount}; }
        fsgpu_bin_batch& v = lib->view;
        v.bases = b.seq.data(); v.quals = b.qua.data(); v.heads = b.head.data(); v.n_bases = b.seq.size(); v.n_heads = b.head.size();
        v.records = lib->records.data(); v.n_records = lib->records.size();
        v.nodes = lib->nodes.data(); v.n_nodes = lib->nodes.size();
        v.top_nodes = b.topNodes.data(); v.n_top_nodes = b.topNodes.size();
        v.em_records = b.emRecs.data(); v.n_em_records = b.emRecs.size();
        v.trees = lib->trees.data(); v.n_trees = lib->trees.size();
        v.bins = lib->bins.data(); v.n_bins = lib->bins.size();
        return lib;
    } catch (const std::exception& e) { g_createError = e.what(); }
    catch (...) { g_createError = "unknown error"; }
    delete lib;
    return nullptr;
}
void fsgpu_library_close(fsgpu_library* lib) { delete lib; }
const fsgpu_bin_batch* fsgpu_library_std_bins(const fsgpu_library* lib) { return lib ? &lib->view : nullptr; }
const void* fsgpu_library_config(const fsgpu_library* lib, size_t* bytes)
{ if (!lib) return nullptr; if (bytes) *bytes = sizeof lib->cfg; return &lib->cfg; }
const uint8_t* fsgpu_library_header_fields(const fsgpu_library* lib, size_t* bytes)
{ if (!lib) return nullptr; if (bytes) *bytes = lib->fields.size(); return lib->fields.empty() ? nullptr : lib->fields.data(); }
const uint8_t* fsgpu_library_quality_codebook(const fsgpu_library* lib, size_t* bytes)
{ if (!lib) return nullptr; if (bytes) *bytes = lib->qvz.size(); return lib->qvz.empty() ? nullptr : lib->qvz.data(); }

int fsgpu_merge_parts(const char* outPrefix, uint32_t world, char* err, size_t errLen)
{
    auto fail = [&](const std::string& m) { if (err && errLen) snprintf(err, errLen, "%s", m.c_str()); return (int)FSGPU_ERR_IO; };
    if (!outPrefix || world == 0) return fail("bad arguments");
    try {
        struct Part { std::vector<uint64_t> sizes; std::vector<uint32_t> sigs; std::vector<uint8_t> rest; std::string data; };
        std::vector<Part> parts(world);
        auto slurp = [](const std::string& name) {
            FILE* f = fopen(name.c_str(), "rb"); if (!f) throw std::runtime_error("Cannot open file: " + name);
            std::vector<uint8_t> v; uint8_t buf[1 << 16]; size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
            fclose(f); return v;
        };
        for (uint32_t r = 0; r < world; ++r) {
            const std::string base = std::string(outPrefix) + ".part" + std::to_string(r);
            const std::vector<uint8_t> m = slurp(base + ".cmeta");
            uint64_t foff = 0, fsize = 0;
            if (m.size() < 24) throw std::runtime_error("Corrupted archive header");
            memcpy(&foff, m.data(), 8); memcpy(&fsize, m.data() + 8, 8);
            if (foff > m.size() || fsize > m.size() - foff || fsize < 4) throw std::runtime_error("Corrupted archive header");
            uint32_t n = 0; memcpy(&n, m.data() + foff, 4);
            if ((uint64_t)n * 12 + 4 > fsize) throw std::runtime_error("Corrupted archive header");
            Part& p = parts[r];
            p.sizes.resize(n); p.sigs.resize(n);
            if (n) { memcpy(p.sizes.data(), m.data() + foff + 4, 8ull * n); memcpy(p.sigs.data(), m.data() + foff + 4 + 8ull * n, 4ull * n); }
            p.rest.assign(m.begin() + (ptrdiff_t)(foff + 4 + 12ull * n), m.begin() + (ptrdiff_t)(foff + fsize));
            p.data = base + ".cdata";
            uint64_t sum = 0; for (uint64_t z : p.sizes) { if (z > (1ull << 40)) throw std::runtime_error("Corrupted archive header"); sum += z; }
            FILE* f = fopen(p.data.c_str(), "rb"); if (!f) throw std::runtime_error("Cannot open file: " + p.data);
            fseeko(f, 0, SEEK_END); const uint64_t have = (uint64_t)ftello(f); fclose(f);
            if (sum != have) throw std::runtime_error("Corrupted archive: " + p.data + " does not hold the blocks its footer lists");
        }
        // global order: rank 0's first block when it is the merged small-bins/N block (signature 4^p), then ascending signature
        struct Ent { uint32_t sig, rank, idx; uint64_t size, off; };
        std::vector<Ent> order, rest;
        auto rawSig = [](uint32_t sg) { return sg > 0 && (sg & (sg - 1)) == 0 && (31 - __builtin_clz(sg)) % 2 == 0; };
        for (uint32_t r = 0; r < world; ++r) {
            uint64_t off = 0;
            for (uint32_t i = 0; i < parts[r].sizes.size(); ++i) {
                const Ent e{parts[r].sigs[i], r, i, parts[r].sizes[i], off}; off += parts[r].sizes[i];
                if (r == 0 && i == 0 && rawSig(e.sig)) order.push_back(e); else rest.push_back(e);
            }
        }
        std::sort(rest.begin(), rest.end(), [](const Ent& a, const Ent& b) { return a.sig != b.sig ? a.sig < b.sig : (a.rank != b.rank ? a.rank < b.rank : a.idx < b.idx); });
        order.insert(order.end(), rest.begin(), rest.end());
        std::vector<FILE*> src(world, nullptr);
        FILE* dst = fopen((std::string(outPrefix) + ".cdata").c_str(), "wb");
        if (!dst) throw std::runtime_error(std::string("Cannot open file: ") + outPrefix + ".cdata");
        std::string problem;
        std::vector<uint8_t> buf;
        for (const Ent& e : order) {
            if (!src[e.rank]) src[e.rank] = fopen(parts[e.rank].data.c_str(), "rb");
            if (!src[e.rank]) { problem = "Cannot open file: " + parts[e.rank].data; break; }
            buf.resize(e.size);
            if (fseeko(src[e.rank], (off_t)e.off, SEEK_SET) != 0 || (e.size && fread(buf.data(), 1, e.size, src[e.rank]) != e.size)) { problem = "Cannot read " + parts[e.rank].data; break; }
            if (e.size && fwrite(buf.data(), 1, e.size, dst) != e.size) { problem = "Cannot write the archive"; break; }
        }
        for (FILE* f : src) if (f) fclose(f);
        if (fclose(dst) != 0 && problem.empty()) problem = "Cannot write the archive";
        if (!problem.empty()) throw std::runtime_error(problem);
        // footer: count, sizes, signatures, then the ArchiveConfig + field table of rank 0's part
        std::vector<uint8_t> body;
        const uint32_t n = (uint32_t)order.size();
        body.resize(4 + 12ull * n);
        memcpy(body.data(), &n, 4);
        for (uint32_t i = 0; i < n; ++i) { memcpy(body.data() + 4 + 8ull * i, &order[i].size, 8); memcpy(body.data() + 4 + 8ull * n + 4ull * i, &order[i].sig, 4); }
        body.insert(body.end(), parts[0].rest.begin(), parts[0].rest.end());
        FILE* cm = fopen((std::string(outPrefix) + ".cmeta").c_str(), "wb");
        if (!cm) throw std::runtime_error(std::string("Cannot open file: ") + outPrefix + ".cmeta");
        uint8_t head[24] = {0}; const uint64_t foff = 24, fsize = body.size();
        memcpy(head, &foff, 8); memcpy(head + 8, &fsize, 8);
        const bool ok = fwrite(head, 1, 24, cm) == 24 && fwrite(body.data(), 1, body.size(), cm) == body.size();
        if (fclose(cm) != 0 || !ok) throw std::runtime_error("Cannot write the archive");
        for (uint32_t r = 0; r < world; ++r) { const std::string base = std::string(outPrefix) + ".part" + std::to_string(r); remove((base + ".cdata").c_str()); remove((base + ".cmeta").c_str()); }
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
    catch (...) { return fail("unknown error"); }
}

int fsgpu_print_stream_sizes(const char* outPrefix, char* err, size_t errLen)
{
    auto fail = [&](const std::string& m) { if (err && errLen) snprintf(err, errLen, "%s", m.c_str()); return (int)FSGPU_ERR_IO; };
    if (!outPrefix) return fail("bad arguments");
    FILE *fm = nullptr, *fd = nullptr;
    try {
        const std::string mname = std::string(outPrefix) + ".cmeta", dname = std::string(outPrefix) + ".cdata";
        fm = fopen(mname.c_str(), "rb"); if (!fm) throw std::runtime_error("Cannot open file: " + mname);
        uint64_t hdr[3] = {0, 0, 0};
        if (fread(hdr, 1, 24, fm) != 24) throw std::runtime_error("Corrupted archive header");
        if (hdr[1] < 4 || hdr[1] > (1ull << 34) || fseeko(fm, (off_t)hdr[0], SEEK_SET) != 0) throw std::runtime_error("Corrupted archive header");
        std::vector<uint8_t> foot(hdr[1]);
        if (fread(foot.data(), 1, foot.size(), fm) != foot.size()) throw std::runtime_error("Corrupted archive header");
        fclose(fm); fm = nullptr;
        uint32_t n = 0; memcpy(&n, foot.data(), 4);
        if (4 + 12ull * n + sizeof(fs::ArchiveConfigRaw) > foot.size()) throw std::runtime_error("Corrupted archive header");
        fs::ArchiveConfigRaw conf; memcpy(&conf, foot.data() + 4 + 12ull * n, sizeof conf);
        if (conf.minParams.signatureLen == 0 || conf.minParams.signatureLen > 15) throw std::runtime_error("Corrupted archive header");
        fs::StreamSizeStats st; st.start(conf.archType, conf.minParams);
        fd = fopen(dname.c_str(), "rb"); if (!fd) throw std::runtime_error("Cannot open file: " + dname);
        std::vector<uint8_t> head(std::max<uint64_t>(st.headerBytes(), 74));
        uint64_t off = 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t size; uint32_t sig; memcpy(&size, foot.data() + 4 + 8ull * i, 8); memcpy(&sig, foot.data() + 4 + 8ull * n + 4ull * i, 4);
            const size_t want = (size_t)std::min<uint64_t>(size, head.size());
            if (fseeko(fd, (off_t)off, SEEK_SET) != 0 || fread(head.data(), 1, want, fd) != want) throw std::runtime_error("Cannot read " + dname);
            st.addBlock(head.data(), size, sig);
            off += size;
        }
        fclose(fd); fd = nullptr;
        st.print(stdout);
        return 0;
    } catch (const std::exception& e) { if (fm) fclose(fm); if (fd) fclose(fd); return fail(e.what()); }
    catch (...) { if (fm) fclose(fm); if (fd) fclose(fd); return fail("unknown error"); }
}

static int encodeStreams(fsgpu_ctx* ctx, size_t n, const uint32_t* kinds, const uint8_t* const* in, const size_t* inLen,
                         uint8_t* const* out, const size_t* outCap, size_t* outLen)
{
    using namespace fsdev;
    FS_GUARD(ctx, {
        std::vector<StreamItem> items(n); uint64_t inBytes = 0;
        for (size_t i = 0; i < n; ++i) {
            StreamItem it; memset(&it, 0, sizeof it);
            const bool rc = kinds[i] != KIND_PPMD;
            const uint64_t bytes = rc ? inLen[i] * 2 : inLen[i];
            if (bytes > 0xF0000000ull) throw std::runtime_error("stream larger than 4 GiB");
            it.kind = kinds[i]; it.in_len = (uint32_t)inLen[i]; it.in_off = inBytes; it.bin = 0;
            it.out_cap = rc ? 2 * it.in_len + 32 : (uint32_t)(bytes + bytes / 8 + 64);
            items[i] = it; inBytes += (bytes + 15) & ~15ull;
        }
        std::vector<uint8_t> input(inBytes + 16);
        for (size_t i = 0; i < n; ++i) { const uint64_t bytes = items[i].kind != KIND_PPMD ? 2ull * items[i].in_len : items[i].in_len; if (bytes) memcpy(input.data() + items[i].in_off, in[i], bytes); }
        // one pseudo-bin per stream so that the assemble step hands every stream back separately
        std::vector<fsdev::BlockPlan> plans; std::vector<uint8_t> blocks; std::vector<uint64_t> sizes;
        std::vector<uint8_t> raw; std::vector<uint32_t> rawSizes;
        if (fsengine::encode_streams_raw(ctx->c.device(), input.data(), inBytes, items, raw, rawSizes, &ctx->c.timing) != 0) throw std::runtime_error(std::string("device: ") + ctx->c.dev->err);
        for (size_t i = 0; i < n; ++i) {
            if (rawSizes[i] == 0xFFFFFFFFu) throw std::runtime_error("stream " + std::to_string(i) + ": symbol or context outside its coder's alphabet");
            outLen[i] = rawSizes[i];
            const size_t c = std::min<size_t>(rawSizes[i], outCap[i]);
            if (c) memcpy(out[i], raw.data() + items[i].out_off, c);
        }
    });
}

int fsgpu_ppmd_encode(fsgpu_ctx* ctx, size_t n, const uint8_t* const* in, const size_t* inLen, uint8_t* const* out, const size_t* outCap, size_t* outLen)
{
    if (!ctx || (n && (!in || !inLen || !out || !outCap || !outLen))) return FSGPU_ERR_ARG;
    std::vector<uint32_t> kinds(n, fsdev::KIND_PPMD);
    return encodeStreams(ctx, n, kinds.data(), in, inLen, out, outCap, outLen);
}

int fsgpu_rc_encode(fsgpu_ctx* ctx, size_t n, const uint32_t* model, const uint8_t* const* pairs, const size_t* nPairs, uint8_t* const* out,
                    const size_t* outCap, size_t* outLen)
{
    if (!ctx || (n && (!model || !pairs || !nPairs || !out || !outCap || !outLen))) return FSGPU_ERR_ARG;
    std::vector<uint32_t> kinds(n);
    for (size_t i = 0; i < n; ++i) { if (model[i] > 6) return FSGPU_ERR_ARG; kinds[i] = fsdev::KIND_RC_BASE + model[i]; }
    return encodeStreams(ctx, n, kinds.data(), pairs, nPairs, out, outCap, outLen);
}

int fsgpu_gather_quality(fsgpu_ctx* ctx, const uint8_t* packed, size_t packedBytes, const fsgpu_quality_string* strings, size_t n, uint8_t* out, size_t outCap, size_t* outLen)
{
    if (!ctx || !outLen || (n && (!packed || !strings || !out))) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        uint64_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            if (strings[i].len > 0xFFFFu || strings[i].src_bit + 6ull * strings[i].len > 8ull * packedBytes) throw std::runtime_error("quality string outside the packed scores");
            total += strings[i].len;
        }
        if (total > outCap || total > 0xF0000000ull || n > 0xFFFFFFF0ull) throw std::runtime_error("output buffer too small for the gathered quality stream");
        *outLen = (size_t)total;
        if (n == 0) return FSGPU_OK;
        // the device input: packed scores (+ slack for the kernel's word reads), then the descriptors
        const uint64_t descOff = ((uint64_t)packedBytes + 8u + 15u) & ~15ull;
        std::vector<uint8_t> input(descOff + n * sizeof(fsdev::QuaString) + 16, 0);
        memcpy(input.data(), packed, packedBytes);
        fsdev::QuaString* qs = (fsdev::QuaString*)(input.data() + descOff);
        uint64_t dst = 0;
        for (size_t i = 0; i < n; ++i) { qs[i].src_bit = strings[i].src_bit; qs[i].dst_off = (uint32_t)dst; qs[i].len = (uint16_t)strings[i].len; qs[i].reverse = strings[i].reverse ? 1 : 0; dst += strings[i].len; }
        fsdev::GatherPlan gp; gp.desc_off = descOff; gp.n_strings = (uint32_t)n; gp.out_bytes = (total + 15u) & ~15ull; gp.symbols = total;
        std::vector<uint8_t> res;
        if (fsengine::gather_quality_raw(ctx->c.device(), input.data(), descOff + n * sizeof(fsdev::QuaString), gp, res, &ctx->c.timing) != 0) throw std::runtime_error(std::string("device: ") + ctx->c.dev->err);
        memcpy(out, res.data(), total);
    });
}

int fsgpu_gather_quality_binned(fsgpu_ctx* ctx, const uint8_t* packed, size_t packedBytes, uint32_t bits, uint32_t binaryThreshold,
                                const fsgpu_quality_string_n* strings, size_t n, uint8_t* out, size_t outCapPairs, size_t* outPairs)
{
    if (!ctx || !outPairs || (bits != 3u && bits != 1u) || (n && (!packed || !strings || !out))) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        uint64_t total = 0, nBytes = 0;
        for (size_t i = 0; i < n; ++i) {
            if (strings[i].len > 255u || strings[i].n_count > strings[i].len || strings[i].src_bit + (uint64_t)bits * strings[i].len > 8ull * packedBytes || (strings[i].n_count && !strings[i].n_positions))
                throw std::runtime_error("quality string outside the packed scores");
            total += strings[i].len - strings[i].n_count; nBytes += strings[i].n_count;
        }
        if (total > outCapPairs || total > 0x70000000ull || n > 0xFFFFFFF0ull) throw std::runtime_error("output buffer too small for the gathered quality stream");
        *outPairs = (size_t)total;
        if (n == 0) return FSGPU_OK;
        const uint64_t descOff = ((uint64_t)packedBytes + 8u + 15u) & ~15ull, nOff = descOff + ((n * sizeof(fsdev::QuaPairString) + 15u) & ~15ull);
        std::vector<uint8_t> input(nOff + nBytes + 32, 0);
        memcpy(input.data(), packed, packedBytes);
        fsdev::QuaPairString* qs = (fsdev::QuaPairString*)(input.data() + descOff);
        uint64_t dst = 0, np = 0;
        for (size_t i = 0; i < n; ++i) {
            qs[i].src_bit = strings[i].src_bit; qs[i].dst_off = (uint32_t)dst; qs[i].n_off = (uint32_t)np; qs[i].len = (uint16_t)strings[i].len; qs[i].reverse = strings[i].reverse ? 1 : 0; qs[i].n_count = (uint8_t)strings[i].n_count;
            if (strings[i].n_count) memcpy(input.data() + nOff + np, strings[i].n_positions, strings[i].n_count);
            dst += strings[i].len - strings[i].n_count; np += strings[i].n_count;
        }
        fsdev::GatherPlan gp; gp.desc_off = descOff; gp.n_strings = (uint32_t)n; gp.out_bytes = (2 * total + 15u) & ~15ull; gp.symbols = total; gp.bits = bits;
        gp.n_list_off = nOff; gp.n_list_bytes = nBytes; gp.sym_of_bit[0] = 6u >= binaryThreshold ? 1u : 0u; gp.sym_of_bit[1] = 40u >= binaryThreshold ? 1u : 0u;
        std::vector<uint8_t> res;
        if (fsengine::gather_quality_raw(ctx->c.device(), input.data(), nOff + nBytes + 16, gp, res, &ctx->c.timing) != 0) throw std::runtime_error(std::string("device: ") + ctx->c.dev->err);
        memcpy(out, res.data(), 2 * total);
    });
}

int fsgpu_tokeniser_check(fsgpu_ctx* ctx, const char* inPrefix, uint64_t* ids, uint64_t* differingBins)
{
    if (!ctx || !inPrefix || !ids || !differingBins) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, ctx->c.tokeniserCheck(inPrefix, *ids, *differingBins));
}

int fsgpu_emit_check(fsgpu_ctx* ctx, const char* inPrefix, uint64_t* ops, uint64_t* streams, uint64_t* differing)
{
    if (!ctx || !inPrefix || !ops || !streams || !differing) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, ctx->c.emitCheck(inPrefix, *ops, *streams, *differing));
}

int fsgpu_pe_matcher_check(fsgpu_ctx* ctx, const char* inPrefix, uint64_t* pairs, uint64_t* differing)
{
    if (!ctx || !inPrefix || !pairs || !differing) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, ctx->c.mateMatcherCheck(inPrefix, *pairs, *differing));
}

int fsgpu_matcher_check(fsgpu_ctx* ctx, const char* inPrefix, uint64_t* reads, uint64_t* differing)
{
    if (!ctx || !inPrefix || !reads || !differing) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, ctx->c.matcherCheck(inPrefix, *reads, *differing));
}

int fsgpu_unpack_check(fsgpu_ctx* ctx, const char* inPrefix, uint64_t* words, uint64_t* differing, uint64_t* reads, uint64_t* differingRows)
{
    if (!ctx || !inPrefix || !words || !differing || !reads || !differingRows) return FSGPU_ERR_ARG;
    *words = *differing = 0;
    if (!fs::Context::deviceUnpack()) { ctx->c.err = "device-side unpack is switched off (FS_DEVICE_UNPACK=0)"; return FSGPU_ERR_ARG; }
    fsengine::unpack_check(true);
    struct Off { ~Off() { fsengine::unpack_check(false); } } off;
    FS_GUARD(ctx, { ctx->c.matcherCheck(inPrefix, *reads, *differingRows); fsengine::unpack_check_counts(words, differing); });
}

// A library of several device batches (more standard-bin bases than one batch holds: tens of millions of reads) is bound by the
// streams of its heaviest bins -- a 40 M-pair library has a 150 M-symbol quality stream, 25 s of one wavefront -- and with the
// batches one after the other everything else waited in line behind them (round 3: 41 s, of which 27 s the first batch,
// profiles/r03_config2_40Mpairs_trace.txt).  Such a library goes through TWO pipelines on the one device: this context packs
// the heaviest bins (one batch), a helper context everything else, at the same time; the blocks are held, the two size
// tables added and every pipeline writes its blocks at their places -- the path of a bin-sharded pack over two ranks, with
// the bins dealt by weight class.  FS_SPLIT_PIPELINES=0: one pipeline as before.
static uint32_t wantsSplit(fsgpu_ctx* ctx, const std::string& in)      // 0: one pipeline; else how many
{
    if (ctx->c.cfg.world_size > 1) return 0;
    uint32_t most = 3;
    if (const char* e = getenv("FS_SPLIT_PIPELINES")) { if (atoi(e) == 0) return 0; if (atoi(e) >= 2) most = std::min(3, atoi(e)); }
    fs::BinFile bf; bf.open(in, ctx->c.par.minBinSize);
    uint64_t bases = 0;
    for (uint32_t sg : bf.stdSignatures()) bases += bf.bins().at(sg).totalRawDnaSize;
    const uint64_t cap = ctx->c.cfg.batch_bases ? ctx->c.cfg.batch_bases : (3072ull << 20);
    if (!(bases > cap && bf.stdSignatures().size() >= 64)) return 0;
    return bases > 2 * cap ? most : 2u;
}

// pipelines: 2 (two device batches) or 3 (more): the context takes the heaviest batch's worth of bins, the first helper the next
// batch's worth (three pipelines), the last helper everything else.  (Measured at 60 M pairs with two: the helper's OWN batches
// ran one after the other, 16.9 + 8.0 + 5.9 + 3.6 + 1.7 s beside the context's 26.7 s.)
// Returns false -- nothing done -- when the device has no room for another pipeline (a helper context brings an arena pool and
// staging buffers of its own): the caller then packs with the one pipeline it has, batch after batch, as round 2 did.
static bool packSplit(fsgpu_ctx* ctx, const std::string& in, const std::string& out, int verbose, uint32_t pipelines)
{
    fs::Context& a = ctx->c;
    const uint32_t T = a.hostThreads;
    // coder lanes: the two matcher streams and all pipelines' lanes share the 16 hardware queues
    const uint32_t lanesOf[2][3] = {{6, 8, 0}, {4, 4, 6}};
    const uint32_t* lanes = lanesOf[pipelines == 3 ? 1 : 0];
    const uint32_t want = pipelines - 1;
    while (ctx->helpers.size() < want) {
        fsgpu_config hc = a.cfg;
        // (a context that packs once has helpers that pack once: they too give their batch's host memory back while the device walks the
        // long streams -- 52 GB resident at the end of a 25 M-pair pack was 1.4 s of teardown, one munmap at a time whatever the threads)
        hc.world_size = 1; hc.rank = 0; hc.host_threads = 1;      // (set for each pack below)
        fsengine::Device* parentDev = nullptr;
        if (!(getenv("FS_SPLIT_POOLS") && atoi(getenv("FS_SPLIT_POOLS")) != 0)) { try { parentDev = a.device(); } catch (...) { parentDev = nullptr; } }      // (FS_SPLIT_POOLS=1: a pool of its own per pipeline, A/B runs)
        t_poolDonor = parentDev;
        fsgpu_ctx* h = fsgpu_create(&hc);
        t_poolDonor = nullptr;
        if (!h) {
            if (verbose > 1 || getenv("FS_TRACE")) fprintf(stderr, "[split] pipeline %zu could not be made (%s): one pipeline\n", ctx->helpers.size() + 2, fsgpu_create_error());
            return false;
        }
        ctx->helpers.push_back(h);
    }
    std::vector<fs::Context*> cs{&a};
    for (uint32_t k = 0; k < want; ++k) cs.push_back(&ctx->helpers[k]->c);
    // (whatever happens, the contexts get their own configuration back; after an error none of them keeps the blocks it held --
    // a library of this size is tens of gigabytes of them)
    struct Keep { std::vector<fs::Context*> cs; std::vector<fsgpu_config> cfg; std::vector<uint32_t> threads; bool done = false;
                  ~Keep() { for (size_t i = 0; i < cs.size(); ++i) { cs[i]->cfg = cfg[i]; cs[i]->hostThreads = threads[i]; cs[i]->splitRole = 0; if (!done) { cs[i]->shards.clear(); cs[i]->shards.shrink_to_fit(); } } } } keep;
    for (fs::Context* c : cs) { keep.cs.push_back(c); keep.cfg.push_back(c->cfg); keep.threads.push_back(c->hostThreads); }
    // Host threads: every pipeline has all of them, and a bin's front end holds one of T worker slots they share.  Who is served first:
    // (three pipelines) the heaviest class, whose front end is the step's critical path (its last slice's streams start when its last bin is
    // through) -- with at most two thirds of the slots; the other third is the LIGHTEST class's from the first millisecond: its many short
    // streams go through the device while it is still empty (behind the heavy classes' resident waves the same streams took 4.3 s instead
    // of 1); the middle class fills what is left.  (Rounds 3-4 and the start of round 5: a quarter of the threads each for the heavy
    // classes of three, a third of two -- the heaviest class's 183 bins of a 25 M-pair library then took six threads 9.0 s while the lightest
    // class's twelve were done after 2.7 s, profiles/r05_cli_trace_pe_25m.txt.  FS_SPLIT_GATE=0: those fixed shares.)
    const bool gated = !(getenv("FS_SPLIT_GATE") && atoi(getenv("FS_SPLIT_GATE")) == 0);
    fs::HostGate gate(T);
    {
        // the heaviest class first, but never with more than two thirds of the slots: the rest is the lightest class's from the start (it
        // takes what the heaviest leaves as well), the middle class of three comes last.  (With the lightest class first and a cap by its share
        // of the bases a 40 M-pair library, whose last class holds half the bases, gave the heaviest class 8 slots: 18.1 -> 19.7 s a step.)
        gate.rank[0] = 0; gate.rank[pipelines - 1u] = 1; if (pipelines == 3) gate.rank[1] = 2;
        gate.cap[0] = std::max(1u, T - std::max(1u, T / 3u));
        if (getenv("FS_TRACE")) fprintf(stderr, "[trace] split pack: %u pipelines, %u worker slots, the heaviest class first with up to %u\n", pipelines, T, gate.cap[0]);
    }
    struct Ungate { std::vector<fs::Context*>& cs; ~Ungate() { for (fs::Context* c : cs) { c->hostGate = nullptr; c->gateClass = 0; c->sliceThreads = 0; } } } ungate{cs};
    for (uint32_t k = 0; k < pipelines; ++k) {
        fs::Context& c = *cs[k];
        c.cfg.rank = k; c.cfg.world_size = pipelines; c.cfg.batch_bases = a.cfg.batch_bases; c.splitRole = k + 1;
        // (the heavy classes: a slice per lane -- measured at 60 M pairs: with two slices per lane the later slices wait for lanes that the
        // first ones hold for the whole of their long streams, 32.6 -> 43.8 s; host threads a quarter each of three, a third of two: with a
        // third each of three the last pipeline, which has most of the bins, becomes the longest, profiles/r03_config2_60Mpairs_*)
        c.cfg.pipeline_lanes = lanes[k]; c.cfg.pipeline_slices = k + 1 < pipelines ? lanes[k] : 0;
        const uint32_t heavy = std::max(1u, pipelines == 3 ? T / 4u : T / 3u);
        const uint32_t share = k + 1 < pipelines ? heavy : std::max(1u, T > heavy * (pipelines - 1) ? T - heavy * (pipelines - 1) : 1u);
        if (gated) { c.hostThreads = T; c.sliceThreads = share; c.hostGate = &gate; c.gateClass = std::min(k, 3u); }
        else c.hostThreads = share;
    }
    std::vector<std::string> errs(pipelines);
    auto all = [&](const std::function<void(uint32_t)>& f) {
        std::vector<std::thread> th;
        for (uint32_t k = 1; k < pipelines; ++k) th.emplace_back([&, k]() { try { f(k); } catch (const std::exception& e) { errs[k] = e.what(); } });
        try { f(0); } catch (const std::exception& e) { errs[0] = e.what(); }
        for (auto& t : th) t.join();
        for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(e);
    };
    all([&](uint32_t k) { cs[k]->shardPack({in}); });
    std::vector<uint32_t> sg; std::vector<uint64_t> sum;
    a.shardTable(0, sg, sum);
    for (uint32_t k = 1; k < pipelines; ++k) {
        std::vector<uint32_t> sg2; std::vector<uint64_t> sz;
        cs[k]->shardTable(0, sg2, sz);
        if (sg != sg2) throw std::runtime_error("the pipelines disagree about the archive's block table");
        for (size_t i = 0; i < sum.size(); ++i) sum[i] += sz[i];
    }
    all([&](uint32_t k) { cs[k]->shardWrite(0, out, sum); });
    // what the helpers did counts as this context's (the callers read one context's statistics)
    for (uint32_t k = 1; k < pipelines; ++k) {
        fs::Context& b = *cs[k];
        a.stats.bins += b.stats.bins; a.stats.records += b.stats.records; a.stats.algorithmic_bytes += b.stats.algorithmic_bytes; a.stats.cdata_bytes += b.stats.cdata_bytes;
        a.stats.device_batches += b.stats.device_batches; a.stats.frontend_ms = std::max(a.stats.frontend_ms, b.stats.frontend_ms);
        a.timing.encode_ms += b.timing.encode_ms; a.timing.assemble_ms += b.timing.assemble_ms; a.timing.launches += b.timing.launches; a.timing.items += b.timing.items;
        a.timing.ppmd_symbols += b.timing.ppmd_symbols; a.timing.rc_symbols += b.timing.rc_symbols; a.timing.restarts += b.timing.restarts; a.timing.max_restarts = std::max(a.timing.max_restarts, b.timing.max_restarts);
        a.timing.h2d_bytes += b.timing.h2d_bytes; a.timing.gather_ms += b.timing.gather_ms; a.timing.gather_symbols += b.timing.gather_symbols; a.timing.gather_bytes += b.timing.gather_bytes; a.timing.id_strings += b.timing.id_strings;
        for (int w = 0; w < 16; ++w) a.timing.win[w] += b.timing.win[w];
        a.timing.tail_launches += b.timing.tail_launches;
        a.matchedReads += b.matchedReads.load(); a.matchUs += b.matchUs.load(); a.matchKernelUs += b.matchKernelUs.load(); a.matchBasesUp += b.matchBasesUp.load(); a.matchUnpackedReads += b.matchUnpackedReads.load();
        a.matedPairs += b.matedPairs.load(); a.mateUs += b.mateUs.load(); a.mateKernelUs += b.mateKernelUs.load(); b.matedPairs = 0; b.mateUs = 0; b.mateKernelUs = 0;
        b.matchedReads = 0; b.matchUs = 0; b.matchKernelUs = 0; b.matchBasesUp = 0; b.matchUnpackedReads = 0;
        b.stats = fsgpu_stats(); b.timing = fsengine::BatchTiming();
    }
    if (verbose) {
        fprintf(stderr, "\rParts processed: %zu (100%%) \n", sum.size());
        char err[256] = {0};
        if (verbose == 1 && fsgpu_print_stream_sizes(out.c_str(), err, sizeof err) != 0) throw std::runtime_error(err);
    }
    keep.done = true;
    return true;
}

int fsgpu_pack_file(fsgpu_ctx* ctx, const char* inPrefix, const char* outPrefix, int verbose)
{
    if (!ctx || !inPrefix || !outPrefix) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        const uint32_t pipelines = wantsSplit(ctx, inPrefix);
        if (!(pipelines && packSplit(ctx, inPrefix, outPrefix, verbose, pipelines))) ctx->c.packFiles({std::string(inPrefix)}, {std::string(outPrefix)}, verbose);
    });
}

int fsgpu_shard_pack(fsgpu_ctx* ctx, const char* inPrefix, size_t* nBlocks)
{
    if (!ctx || !inPrefix || !nBlocks) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, { ctx->c.shardPack({std::string(inPrefix)}); *nBlocks = ctx->c.shards.at(0).order.size(); });
}

int fsgpu_shard_table(const fsgpu_ctx* ctx, uint32_t* signatures, uint64_t* sizes, size_t nBlocks)
{
    if (!ctx || !signatures || !sizes) return FSGPU_ERR_ARG;
    fsgpu_ctx* c = const_cast<fsgpu_ctx*>(ctx);
    FS_GUARD(c, {
        std::vector<uint32_t> sg; std::vector<uint64_t> sz;
        ctx->c.shardTable(0, sg, sz);
        if (sg.size() != nBlocks) throw std::runtime_error("block table size mismatch");
        std::copy(sg.begin(), sg.end(), signatures); std::copy(sz.begin(), sz.end(), sizes);
    });
}

int fsgpu_shard_write(fsgpu_ctx* ctx, const char* outPrefix, const uint64_t* allSizes, size_t nBlocks)
{
    if (!ctx || !outPrefix || !allSizes) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, ctx->c.shardWrite(0, outPrefix, std::vector<uint64_t>(allSizes, allSizes + nBlocks)));
}

// the same three steps for a SET of libraries packed in one device pipeline (lib = index into in_prefixes)
int fsgpu_shard_pack_set(fsgpu_ctx* ctx, size_t n, const char* const* inPrefixes, size_t* nBlocks)
{
    if (!ctx || !n || !inPrefixes || !nBlocks) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        std::vector<std::string> in; for (size_t i = 0; i < n; ++i) { if (!inPrefixes[i]) throw std::runtime_error("null prefix"); in.emplace_back(inPrefixes[i]); }
        ctx->c.shardPack(in);
        for (size_t i = 0; i < n; ++i) nBlocks[i] = ctx->c.shards.at(i).order.size();
    });
}

int fsgpu_shard_table_of(const fsgpu_ctx* ctx, size_t lib, uint32_t* signatures, uint64_t* sizes, size_t nBlocks)
{
    if (!ctx || !signatures || !sizes) return FSGPU_ERR_ARG;
    fsgpu_ctx* c = const_cast<fsgpu_ctx*>(ctx);
    FS_GUARD(c, {
        std::vector<uint32_t> sg; std::vector<uint64_t> sz;
        ctx->c.shardTable(lib, sg, sz);
        if (sg.size() != nBlocks) throw std::runtime_error("block table size mismatch");
        std::copy(sg.begin(), sg.end(), signatures); std::copy(sz.begin(), sz.end(), sizes);
    });
}

int fsgpu_shard_write_of(fsgpu_ctx* ctx, size_t lib, const char* outPrefix, const uint64_t* allSizes, size_t nBlocks)
{
    if (!ctx || !outPrefix || !allSizes) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, ctx->c.shardWrite(lib, outPrefix, std::vector<uint64_t>(allSizes, allSizes + nBlocks)));
}

int fsgpu_pack_files(fsgpu_ctx* ctx, size_t n, const char* const* inPrefixes, const char* const* outPrefixes, int verbose)
{
    if (!ctx || !n || !inPrefixes || !outPrefixes) return FSGPU_ERR_ARG;
    FS_GUARD(ctx, {
        std::vector<std::string> a, b;
        for (size_t i = 0; i < n; ++i) { a.emplace_back(inPrefixes[i]); b.emplace_back(outPrefixes[i]); }
        ctx->c.packFiles(a, b, verbose);
    });
}

int fsgpu_reset_stats(fsgpu_ctx* ctx)
{
    if (!ctx) return FSGPU_ERR_ARG;
    ctx->c.stats = fsgpu_stats(); ctx->c.timing = fsengine::BatchTiming(); ctx->c.matchedReads = 0; ctx->c.matchUs = 0; ctx->c.matchKernelUs = 0; ctx->c.matchBasesUp = 0; ctx->c.matchUnpackedReads = 0;
    ctx->c.matedPairs = 0; ctx->c.mateUs = 0; ctx->c.mateKernelUs = 0;
    return FSGPU_OK;
}

int fsgpu_get_stats(const fsgpu_ctx* ctx, fsgpu_stats* out)
{
    if (!ctx || !out) return FSGPU_ERR_ARG;
    *out = ctx->c.stats;
    out->encode_kernel_ms = ctx->c.timing.encode_ms; out->assemble_kernel_ms = ctx->c.timing.assemble_ms;
    out->kernel_launches = ctx->c.timing.launches; out->stream_items = ctx->c.timing.items; out->ppmd_symbols = ctx->c.timing.ppmd_symbols;
    out->ppmd_window_attempts = ctx->c.timing.win[1]; out->ppmd_windows = ctx->c.timing.win[2]; out->ppmd_window_symbols = ctx->c.timing.win[3];
    out->ppmd_window_rounds = ctx->c.timing.win[4]; out->ppmd_windows_redone = ctx->c.timing.win[5]; out->ppmd_window_light_rounds = ctx->c.timing.win[6];
    out->tokenised_ids = ctx->c.timing.id_strings; out->matcher_reads = ctx->c.matchedReads.load(); out->matcher_call_ms = ctx->c.matchUs.load() / 1e3; out->matcher_kernel_ms = ctx->c.matchKernelUs.load() / 1e3;
    out->matcher_bases_h2d_bytes = ctx->c.matchBasesUp.load(); out->matcher_unpacked_reads = ctx->c.matchUnpackedReads.load();
    out->mate_pairs = ctx->c.matedPairs.load(); out->mate_call_ms = ctx->c.mateUs.load() / 1e3; out->mate_kernel_ms = ctx->c.mateKernelUs.load() / 1e3;
    out->gather_kernel_ms = ctx->c.timing.gather_ms; out->gather_symbols = ctx->c.timing.gather_symbols; out->gather_bytes = ctx->c.timing.gather_bytes;
    out->rc_symbols = ctx->c.timing.rc_symbols; out->ppmd_restarts = ctx->c.timing.restarts; out->ppmd_max_restarts = ctx->c.timing.max_restarts;
    out->h2d_bytes = ctx->c.timing.h2d_bytes; out->d2h_bytes = ctx->c.timing.d2h_bytes;
    out->struct_bytes = sizeof(fsgpu_stats); out->ppmd_window_drops = ctx->c.timing.win[7]; out->coder_tail_launches = ctx->c.timing.tail_launches;
    return FSGPU_OK;
}

int fsgpu_get_window_profile(const fsgpu_ctx* ctx, uint64_t out[8])
{
    if (!ctx || !out) return FSGPU_ERR_ARG;
    for (int i = 0; i < 8; ++i) out[i] = ctx->c.timing.win[8 + i];
    return FSGPU_OK;
}

int fsgpu_get_serial_profile(const fsgpu_ctx* ctx, uint64_t out[2])
{
    if (!ctx || !out) return FSGPU_ERR_ARG;
    out[0] = ctx->c.timing.win[6]; out[1] = ctx->c.timing.win[7];
    return FSGPU_OK;
}

}  // extern "C"
