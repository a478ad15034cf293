// TEST INFRASTRUCTURE (oracle) -- not part of the product path.
//
// Thin command-line harness around the *unmodified* reference sources under
// /root/reference/fastore (compiled where they lie by oracle/Makefile; outputs go to
// oracle/_ref/).  It replaces the three reference main.cpp files (which need a
// Makefile-generated version.cpp) and calls the reference's module classes directly:
//   bin   -> BinModuleSE/PE::Fastq2Bin           (fastore_bin/BinModule.h:25-51)
//   rebin -> RebinModule::Bin2Bin                (fastore_rebin/RebinModule.h:25-36)
//   pack  -> CompressorModuleSE/PE::Bin2Dnarch   (fastore_pack/CompressorModule.h:29-49)
//   unpack-> CompressorModuleSE/PE::Dnarch2Dna
// Flag letters follow the reference CLIs (fastore_bin/main.cpp:166-265,
// fastore_rebin/main.cpp:113-180, fastore_pack/main.cpp:165-301) so the C1 profile of
// scripts/fastore_compress.sh:146-148 can be passed through verbatim.
//
//   ppmd  -> PpmdEncoder::StartCompress(4,16) + EncodeNextMember, exactly as
//            LzCompressorSE does (fastore_pack/FastqCompressor.cpp:772-774, 1096-1118)
//   rc    -> one of the range-coder context models of FastqCompressor.h:159-160,250-251,
//            478-480,1076 driven with (symbol, ctx0) byte pairs: golden vectors for the
//            oracle's range-coder restatement
//
//   qvz   -> the reference's --lossy quality path on a list of reads: QvzCodebook::ReadCodebook
//            (fastore_bin/QVZ.cpp:225-302) on the .bmeta footer section, then per read the loop of
//            IQualityStoreBase::CompressReadQuality (fastore_pack/FastqCompressor.cpp:318-364) --
//            choose_quantizer with the WELL generator, quantize, QVZEncoder::EncodeNext -- and End():
//            golden vectors for the oracle's QVZ restatement
//
// usage: ref_driver <bin|rebin|pack|unpack> [flags]
//        ref_driver qvz <footer: WELL state, max_read_length, codebook> <reads: u32 n, u32 len[n], quality values> <out>
//        ref_driver ppmd <in> <out>          ref_driver ppmdd <in> <out>   (PpmdDecoder: inverse, for debugging)
//        ref_driver rc <model> <in: sym,ctx byte pairs> <out>     model = s2o4|s8o4|a8o4|a2o10|a8o6|a256o1

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "fastore_bin/Globals.h"
#include "fastore_bin/BinModule.h"
#include "fastore_bin/Params.h"
#include "fastore_rebin/RebinModule.h"
#include "fastore_rebin/Params.h"
#include "fastore_pack/CompressorModule.h"
#include "fastore_pack/Params.h"
#include "fastore_pack/distortion.h"
#include "fastore_bin/Buffer.h"
#include "fastore_bin/BitMemory.h"
#include "rc/ContextEncoder.h"
#include "ppmd/PPMd.h"
#include "fastore_bin/QVZ.h"
#include "fastore_pack/qv_compressor.h"

static std::vector<std::string> split_ws(const char* s)
{
    std::vector<std::string> out;
    std::string cur;
    for (; *s; ++s) {
        if (*s == ' ' || *s == '\n') { if (!cur.empty()) out.push_back(cur); cur.clear(); }
        else cur.push_back(*s);
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

static int num(const char* p)
{
    size_t n = strlen(p);
    if (n == 0 || n >= 8) return -1;
    return atoi(p);
}

static int do_bin(int argc, char** argv)
{
    BinModuleConfig cfg;
    cfg.binningType = BinModuleConfig::BIN_RECORDS;
    memset(&cfg.quaParams.qvzOpts, 0, sizeof(cfg.quaParams.qvzOpts));
    cfg.quaParams.qvzOpts.distortion = DISTORTION_MSE;
    cfg.quaParams.qvzOpts.D = 1;
    std::vector<std::string> in, out;
    unsigned threads = 1; bool verbose = false, gz = false;
    for (int i = 2; i < argc; ++i) {
        const char* p = argv[i];
        if (p[0] != '-') continue;
        int v = num(p + 2);
        switch (p[1]) {
        case 'i': in = split_ws(p + 2); break;
        case 'o': out = split_ws(p + 2); break;
        case 'g': gz = true; break;
        case 'b': cfg.fastqBlockSize = (uint64)v << 20; break;
        case 't': threads = v; break;
        case 'v': verbose = true; break;
        case 'z': cfg.archiveType.readType = ArchiveType::READ_PE; break;
        case 'p': cfg.minimizer.signatureLen = v; break;
        case 's': cfg.minimizer.skipZoneLen = v; break;
        case 'm': cfg.catParams.minBlockBinSize = v; break;
        case 'H': cfg.archiveType.readsHaveHeaders = true; break;
        case 'C': cfg.headParams.preserveComments = false; break;
        case 'q': cfg.quaParams.method = v; break;
        case 'w': cfg.quaParams.binaryThreshold = v; break;
        case 'I': cfg.archiveType.qualityOffset = ArchiveType::Illumina64QualityOffset; break;
        case 'T': cfg.quaParams.qvzOpts.D = atof(p + 2); break;
        }
    }
    if (in.empty() || out.empty()) { fprintf(stderr, "bin: need -i and -o\n"); return 2; }
    if (cfg.archiveType.readType == ArchiveType::READ_PE) {
        if (in.size() % 2) { fprintf(stderr, "bin: PE needs an even number of inputs\n"); return 2; }
        std::vector<std::string> f1(in.begin(), in.begin() + in.size() / 2), f2(in.begin() + in.size() / 2, in.end());
        BinModulePE m; m.Fastq2Bin(f1, f2, out[0], cfg, threads, gz, verbose);
    } else {
        BinModuleSE m; m.Fastq2Bin(in, out[0], cfg, threads, gz, verbose);
    }
    return 0;
}

static int do_rebin(int argc, char** argv)
{
    BinBalanceParameters par;
    std::vector<std::string> in, out;
    unsigned threads = 1; bool verbose = false;
    for (int i = 2; i < argc; ++i) {
        const char* p = argv[i];
        if (p[0] != '-') continue;
        int v = num(p + 2);
        switch (p[1]) {
        case 'i': in = split_ws(p + 2); break;
        case 'o': out = split_ws(p + 2); break;
        case 'p': par.signatureParity = v; break;
        case 'x': par.minBinSizeToExtract = v; break;
        case 'y': par.minBinSizeToCategorize = v; break;
        case 'q': par.minTreeSize = v; break;
        case 'e': par.classifier.encodeThresholdValue = v; break;
        case 's': par.classifier.shiftCost = v; break;
        case 'm': par.classifier.mismatchCost = v; break;
        case 'w': par.classifier.maxLzWindowSize = v; break;
        case 'r': par.classifier.extraReduceHardReads = true; break;
        case 'l': par.classifier.extraReduceExpensiveLzMatches = true; break;
        case 't': threads = v; break;
        case 'v': verbose = true; break;
        case 'z': break;   // PE is read from the .bmeta config
        }
    }
    if (in.empty() || out.empty()) { fprintf(stderr, "rebin: need -i and -o\n"); return 2; }
    RebinModule m; m.Bin2Bin(in[0], out[0], par, threads, verbose);
    return 0;
}

static int do_pack(int argc, char** argv, bool decode)
{
    CompressorParams par;
    CompressorAuxParams aux;
    std::string in; std::vector<std::string> out;
    unsigned threads = 1; bool verbose = false, pe = false;
    for (int i = 2; i < argc; ++i) {
        const char* p = argv[i];
        if (p[0] != '-') continue;
        int v = num(p + 2);
        switch (p[1]) {
        case 'i': in = p + 2; break;
        case 'o': out = split_ws(p + 2); break;
        case 't': threads = v; break;
        case 'v': verbose = true; break;
        case 'z': pe = true; break;
        case 'f': par.extractor.minBinSize = v; break;
        case 'w': par.classifier.maxLzWindowSize = v; break;
        case 'W': par.classifier.maxPairLzWindowSize = v; break;
        case 'e': par.classifier.encodeThresholdValue = v; break;
        case 'E': par.classifier.pairEncodeThresholdValue = v; break;
        case 's': par.classifier.shiftCost = v; break;
        case 'm': par.classifier.mismatchCost = v; break;
        case 'r': par.classifier.extraReduceHardReads = true; break;
        case 'l': par.classifier.extraReduceExpensiveLzMatches = true; break;
        case 'q': par.consensus.maxRecordShiftDifference = v; break;
        case 'n': par.consensus.maxNewVariantsPerRead = v; break;
        case 'd': par.consensus.maxHammingDistance = v; break;
        case 'c': par.consensus.minConsensusSize = v; break;
        }
    }
    if (in.empty() || out.empty()) { fprintf(stderr, "pack: need -i and -o\n"); return 2; }
    if (!decode) {
        if (pe) { CompressorModulePE m; m.Bin2Dnarch(in, out[0], par, aux, threads, verbose); }
        else    { CompressorModuleSE m; m.Bin2Dnarch(in, out[0], par, aux, threads, verbose); }
    } else {
        if (pe) {
            if (out.size() != 2) { fprintf(stderr, "unpack: PE needs two outputs\n"); return 2; }
            CompressorModulePE m; m.Dnarch2Dna(in, out[0], out[1], threads);
        } else { CompressorModuleSE m; m.Dnarch2Dna(in, out[0], threads); }
    }
    return 0;
}

static bool read_file(const char* fn, std::vector<unsigned char>& v)
{
    FILE* f = fopen(fn, "rb"); if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize(n); if (n && fread(v.data(), 1, n, f) != (size_t)n) { fclose(f); return false; }
    fclose(f); return true;
}
static bool write_file(const char* fn, const unsigned char* p, size_t n)
{
    FILE* f = fopen(fn, "wb"); if (!f) return false;
    if (n) fwrite(p, 1, n, f); fclose(f); return true;
}

static int do_ppmd(int argc, char** argv)
{
    if (argc != 4) return 2;
    std::vector<unsigned char> in; if (!read_file(argv[2], in) || in.empty()) return 1;
    std::vector<unsigned char> out(in.size() * 2 + 4096);
    PpmdEncoder enc; enc.StartCompress(4, 16);
    uint64_t outSize = out.size();
    bool ok = enc.EncodeNextMember(in.data(), in.size(), out.data(), outSize);
    enc.FinishCompress();
    if (!ok) return 1;
    return write_file(argv[3], out.data(), outSize) ? 0 : 1;
}

static void rc_put(TEncoder<TSimpleContextCoder<2, 4>>& x, unsigned s, unsigned) { x.coder.EncodeSymbol(x.rc, s); }
static void rc_put(TEncoder<TSimpleContextCoder<8, 4>>& x, unsigned s, unsigned) { x.coder.EncodeSymbol(x.rc, s); }
template <class X> static void rc_put(X& x, unsigned s, unsigned c) { x.coder.EncodeSymbol(x.rc, s, c); }

static int do_ppmdd(int argc, char** argv)
{
    if (argc != 4) return 2;
    std::vector<unsigned char> in; if (!read_file(argv[2], in) || in.empty()) return 1;
    std::vector<unsigned char> out(in.size() * 64 + (1 << 20));
    PpmdDecoder dec; dec.StartDecompress(16);
    uint64_t outSize = out.size();
    bool ok = dec.DecodeNextMember(in.data(), in.size(), out.data(), outSize);
    dec.FinishDecompress();
    if (!ok) return 1;
    return write_file(argv[3], out.data(), outSize) ? 0 : 1;
}

template <class TCoder, bool kCtx> static int run_rc(const std::vector<unsigned char>& in, const char* outFn)
{
    Buffer buf(1 << 16);
    BitMemoryWriter w(buf);
    TEncoder<TCoder>* e = new TEncoder<TCoder>(w);   // 32 MiB tables for a256o1: heap, like the reference
    e->Start();
    for (size_t i = 0; i + 1 < in.size(); i += 2)
        rc_put(*e, in[i], in[i + 1]);
    e->End();
    w.FlushPartialWordBuffer();
    bool ok = write_file(outFn, w.Pointer(), w.Position());
    delete e;
    return ok ? 0 : 1;
}

static int do_rc(int argc, char** argv)
{
    if (argc != 5) return 2;
    std::vector<unsigned char> in; if (!read_file(argv[3], in)) return 1;
    std::string m = argv[2];
    if (m == "s2o4")   return run_rc<TSimpleContextCoder<2, 4>, false>(in, argv[4]);
    if (m == "s8o4")   return run_rc<TSimpleContextCoder<8, 4>, false>(in, argv[4]);
    if (m == "a8o4")   return run_rc<TAdvancedContextCoder<8, 4>, true>(in, argv[4]);
    if (m == "a2o10")  return run_rc<TAdvancedContextCoder<2, 10>, true>(in, argv[4]);
    if (m == "a8o6")   return run_rc<TAdvancedContextCoder<8, 6>, true>(in, argv[4]);
    if (m == "a256o1") return run_rc<TAdvancedContextCoder<256, 1>, true>(in, argv[4]);
    return 2;
}

static std::vector<unsigned char> slurp(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    std::vector<unsigned char> v; unsigned char buf[65536]; size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

static int do_qvz(int argc, char** argv)
{
    if (argc != 5) { fprintf(stderr, "usage: ref_driver qvz <footer> <reads> <out>\n"); return 2; }
    std::vector<unsigned char> foot = slurp(argv[2]), reads = slurp(argv[3]);
    QualityCompressionData qd;
    {
        Buffer mem(foot.size() + 16);
        memcpy(mem.Pointer(), foot.data(), foot.size());
        BitMemoryReader reader(mem, foot.size());
        reader.GetBytes((byte*)qd.well.state, sizeof(qd.well.state));
        reader.GetBytes((byte*)&qd.max_read_length, sizeof(qd.max_read_length));
        struct alphabet_t* A = alloc_alphabet(ALPHABET_SIZE);
        qd.codebook.ReadCodebook(reader, A, qd.max_read_length);
    }
    uint32_t n; memcpy(&n, reads.data(), 4);
    const uint32_t* lens = (const uint32_t*)(reads.data() + 4);
    const unsigned char* q = reads.data() + 4 + 4ull * n;
    uint64_t total = 0; for (uint32_t i = 0; i < n; ++i) total += lens[i];
    Buffer outBuf(3 * total + 1024);
    BitMemoryWriter writer(outBuf);
    struct cond_quantizer_list_t* qlist = qd.codebook.qlist;
    QVZEncoder* enc = new QVZEncoder(&writer, qlist);
    enc->Start();
    well_state_t well; memset(&well, 0, sizeof well);
    memcpy(well.state, qd.well.state, sizeof well.state);
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t idx = 0, prev = 0;
        for (uint32_t i = 0; i < lens[r]; ++i) {
            struct quantizer_t* qz = choose_quantizer(qlist, &well, i, prev, &idx);
            const uint32_t hat = qz->q[q[i]];
            enc->EncodeNext(get_symbol_index(qz->output_alphabet, hat), i, idx);
            prev = hat;
        }
        q += lens[r];
    }
    enc->End();
    delete enc;
    writer.Flush();
    FILE* f = fopen(argv[4], "wb");
    fwrite(outBuf.Pointer(), 1, writer.Position(), f);
    fclose(f);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: ref_driver <bin|rebin|pack|unpack> [reference flags]\n");
        return 2;
    }
    try {
        std::string cmd = argv[1];
        if (cmd == "bin") return do_bin(argc, argv);
        if (cmd == "rebin") return do_rebin(argc, argv);
        if (cmd == "pack") return do_pack(argc, argv, false);
        if (cmd == "unpack") return do_pack(argc, argv, true);
        if (cmd == "ppmd") return do_ppmd(argc, argv);
        if (cmd == "ppmdd") return do_ppmdd(argc, argv);
        if (cmd == "rc") return do_rc(argc, argv);
        if (cmd == "qvz") return do_qvz(argc, argv);
        fprintf(stderr, "unknown command %s\n", argv[1]);
        return 2;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 255;
    }
}
