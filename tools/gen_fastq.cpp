// Synthetic FASTQ generator for the fastore_pack benchmark / parity inputs.
//
//   gen_fastq --reads N --len L [--paired] --genome G --seed S --out PREFIX
//
// writes PREFIX_1.fastq (and PREFIX_2.fastq with --paired).  PRNG: xoshiro256** seeded via
// splitmix64(S).  Genome: G uniform ACGT bases.  Each read/fragment start is uniform, strand
// 50/50; PE fragment length uniform in [2L, 3L], mate 2 = reverse complement of the fragment
// end.  Per base: substitution 0.5 %, N 0.05 %.  Quality: bounded random walk on [2, 40] from
// 38 with steps {-3,-1,0,0,0,0,+1,+1}, Phred+33.  Header "@SYN.<i> <i>/1" (and "/2").
// --noisy-quality (tests only): every score uniform on [2, 40] instead -- a PPMd model of such a stream outgrows its
// 16 MiB heap about every 1.4 M symbols and restarts (ppmd/Model.cpp:109-140), which the random walk never does.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint64_t s[4];
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t next_u64()
{
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
}
static void seed_rng(uint64_t x)
{
    for (int i = 0; i < 4; ++i) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        s[i] = z ^ (z >> 31);
    }
}
static inline uint64_t bounded(uint64_t n) { return (uint64_t)(((__uint128_t)next_u64() * n) >> 64); }

static bool g_noisyQuality = false;
static const char ACGT[4] = {'A', 'C', 'G', 'T'};
static inline char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; } return 'N'; }

static void emit(FILE* f, uint64_t idx, int mate, const char* frag, int L, bool rc, std::vector<char>& buf)
{
    static const int steps[8] = {-3, -1, 0, 0, 0, 0, 1, 1};
    int n = snprintf(buf.data(), 64, "@SYN.%llu %llu/%d\n", (unsigned long long)idx, (unsigned long long)idx, mate);
    char* p = buf.data() + n;
    for (int i = 0; i < L; ++i) {
        char c = rc ? comp(frag[L - 1 - i]) : frag[i];
        uint64_t r = bounded(10000);
        if (r < 5) c = 'N';
        else if (r < 55) { char d; do d = ACGT[bounded(4)]; while (d == c); c = d; }
        *p++ = c;
    }
    *p++ = '\n'; *p++ = '+'; *p++ = '\n';
    int q = 38;
    for (int i = 0; i < L; ++i) {
        if (g_noisyQuality) q = 2 + (int)bounded(39);
        else { q += steps[next_u64() >> 61]; if (q < 2) q = 2; if (q > 40) q = 40; }
        *p++ = (char)(33 + q);
    }
    *p++ = '\n';
    fwrite(buf.data(), 1, p - buf.data(), f);
}

int main(int argc, char** argv)
{
    uint64_t N = 0, G = 0, S = 1; int L = 100; bool paired = false; std::string out;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--reads") N = strtoull(val(), 0, 10);
        else if (a == "--len") L = atoi(val());
        else if (a == "--genome") G = strtoull(val(), 0, 10);
        else if (a == "--seed") S = strtoull(val(), 0, 10);
        else if (a == "--out") out = val();
        else if (a == "--paired") paired = true;
        else if (a == "--noisy-quality") g_noisyQuality = true;
        else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
    }
    if (!N || !G || out.empty() || L < 20 || L > 250 || G < (uint64_t)(3 * L + 1)) {
        fprintf(stderr, "usage: gen_fastq --reads N --len L [--paired] [--noisy-quality] --genome G --seed S --out PREFIX\n");
        return 2;
    }
    seed_rng(S);
    std::vector<char> genome(G);
    for (uint64_t i = 0; i < G; ++i) genome[i] = ACGT[next_u64() >> 62];
    FILE* f1 = fopen((out + "_1.fastq").c_str(), "wb");
    FILE* f2 = paired ? fopen((out + "_2.fastq").c_str(), "wb") : nullptr;
    if (!f1 || (paired && !f2)) { perror("fopen"); return 1; }
    static char b1[1 << 20], b2[1 << 20];
    setvbuf(f1, b1, _IOFBF, sizeof b1);
    if (f2) setvbuf(f2, b2, _IOFBF, sizeof b2);
    std::vector<char> buf(2 * L + 128), frag(3 * L + 1);
    for (uint64_t i = 1; i <= N; ++i) {
        const int flen = paired ? 2 * L + (int)bounded(L + 1) : L;
        const uint64_t start = bounded(G - flen + 1);
        const bool strand = next_u64() >> 63;
        // fragment in read orientation
        for (int k = 0; k < flen; ++k)
            frag[k] = strand ? comp(genome[start + flen - 1 - k]) : genome[start + k];
        emit(f1, i, 1, frag.data(), L, false, buf);
        if (paired) emit(f2, i, 2, frag.data() + flen - L, L, true, buf);
    }
    fclose(f1); if (f2) fclose(f2);
    return 0;
}
