// TEST-ONLY: see simt.h.  Fibers with a hand-written x86-64 context switch (callee-saved registers + stack pointer).
#include "simt.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <vector>

#if !defined(__x86_64__)
#error "the SIMT emulation's context switch is written for x86-64"
#endif

extern "C" void simt_switch(void** saveSp, void* loadSp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size simt_switch,.-simt_switch
)");

namespace simt {

namespace {
struct Wave {
    void* sp[WAVE];
    bool alive[WAVE];
    int live = WAVE;
    int resume = 0;                 // lane that runs next when this wave gets its turn
    uint64_t slot[2][WAVE];
    uint32_t opCount[WAVE];
};
struct State {
    std::vector<Wave> waves;
    void* mainSp = nullptr;
    uint8_t* stacks = nullptr;
    const std::function<void(int, int)>* body = nullptr;
    int curWave = 0, curLane = 0, liveWaves = 0;
};
thread_local State* g = nullptr;
constexpr size_t kStack = 512 << 10;

// the running fiber gives way: to the next lane of its wave, or -- when every lane of the wave has arrived -- to the
// next wave that still has lanes.  `dying`: the caller has finished and never comes back.
void yield(bool dying)
{
    State& s = *g;
    const int w = s.curWave, me = s.curLane;
    Wave& W = s.waves[w];
    void* dead; void** save = dying ? &dead : &W.sp[me];
    if (dying) { W.alive[me] = false; if (--W.live == 0) --s.liveWaves; }
    if (s.liveWaves == 0) { simt_switch(save, s.mainSp); return; }
    int nw = w, nl = -1;
    if (W.live > 0) {
        int n = me;
        do { n = (n + 1) % WAVE; } while (!W.alive[n]);
        const bool wrapped = n <= me;                 // every live lane of this wave has been through this point
        if (!wrapped) nl = n;
        else { W.resume = n; }
    }
    if (nl < 0) {                                     // hand over to the next wave with live lanes (possibly this one again)
        do { nw = (nw + 1) % (int)s.waves.size(); } while (s.waves[nw].live == 0);
        nl = s.waves[nw].resume;
    }
    if (nw == w && nl == me && !dying) return;
    s.curWave = nw; s.curLane = nl;
    simt_switch(save, s.waves[nw].sp[nl]);
}

void trampoline()
{
    State& s = *g;
    (*s.body)(s.curWave, s.curLane);
    yield(true);
    abort();
}
}  // namespace

void run_waves(int waves, const std::function<void(int, int)>& body)
{
    State st; st.waves.resize(waves);
    const size_t fibers = (size_t)waves * WAVE;
    st.stacks = (uint8_t*)mmap(nullptr, kStack * fibers, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (st.stacks == (uint8_t*)MAP_FAILED) { perror("simt: mmap"); abort(); }
    st.body = &body; st.liveWaves = waves;
    for (int w = 0; w < waves; ++w) {
        Wave& W = st.waves[w];
        memset(W.opCount, 0, sizeof W.opCount);
        for (int l = 0; l < WAVE; ++l) {
            // initial frame: six callee-saved registers, then the return address; the stack is 16-byte aligned at the
            // trampoline's first instruction as after a call (rsp % 16 == 8)
            uint64_t* top = (uint64_t*)(st.stacks + kStack * ((size_t)w * WAVE + l + 1));
            top -= 1;                                   // alignment slot
            *--top = (uint64_t)(uintptr_t)&trampoline;  // ret target
            for (int k = 0; k < 6; ++k) *--top = 0;
            W.sp[l] = top; W.alive[l] = true;
        }
    }
    State* prev = g; g = &st;
    st.curWave = 0; st.curLane = 0;
    simt_switch(&st.mainSp, st.waves[0].sp[0]);
    g = prev;
    munmap(st.stacks, kStack * fibers);
}

void run(const std::function<void(int)>& body) { run_waves(1, [&](int, int lane) { body(lane); }); }

int lane() { return g->curLane; }
int wave() { return g->curWave; }
void barrier() { yield(false); }

// every primitive: publish, meet, read.  The slots alternate between two sets so that a lane that runs ahead into the
// next primitive cannot overwrite a value a slower lane has still to read.
static inline uint64_t* publish(uint64_t v)
{
    State& s = *g;
    Wave& W = s.waves[s.curWave];
    const int me = s.curLane;
    uint64_t* set = W.slot[W.opCount[me]++ & 1u];
    set[me] = v;
    yield(false);
    return set;
}

uint64_t ballot(bool p)
{
    const uint64_t* set = publish(p ? 1u : 0u);
    uint64_t m = 0;
    for (int l = 0; l < WAVE; ++l) m |= (set[l] & 1u) << l;
    return m;
}
uint32_t readlane(uint32_t v, uint32_t srcLane) { const uint64_t* set = publish(v); return (uint32_t)set[srcLane & (WAVE - 1)]; }
uint32_t bperm(uint32_t v, uint32_t srcLane) { const uint64_t* set = publish(v); return (uint32_t)set[srcLane & (WAVE - 1)]; }
uint32_t sum(uint32_t v, bool pred)
{
    const uint64_t* set = publish(pred ? v : 0u);
    uint32_t t = 0;
    for (int l = 0; l < WAVE; ++l) t += (uint32_t)set[l];
    return t;
}
uint32_t readfirst(uint32_t v)
{
    const uint64_t* set = publish(v);
    for (int l = 1; l < WAVE; ++l)
        if ((uint32_t)set[l] != (uint32_t)set[0]) { fprintf(stderr, "simt: value asserted wave-uniform differs between lanes 0 (%u) and %d (%u)\n", (uint32_t)set[0], l, (uint32_t)set[l]); abort(); }
    return (uint32_t)set[0];
}

}  // namespace simt
