// TEST-ONLY: see simt.h.  Fibers with a hand-written x86-64 context switch (callee-saved registers + stack pointer).
#include "simt.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

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
struct State {
    void* sp[WAVE];
    void* mainSp = nullptr;
    uint8_t* stacks = nullptr;
    const std::function<void(int)>* body = nullptr;
    int cur = 0, live = 0;
    uint64_t slot[2][WAVE];
    uint32_t opCount[WAVE];
};
thread_local State* g = nullptr;
constexpr size_t kStack = 512 << 10;

void next()
{
    State& s = *g;
    const int me = s.cur;
    if (s.live == 0) { simt_switch(&s.sp[me], s.mainSp); return; }
    int n = me;
    do { n = (n + 1) % WAVE; } while (s.sp[n] == nullptr && n != me);      // finished lanes have no context any more
    if (n == me) return;
    s.cur = n;
    simt_switch(&s.sp[me], s.sp[n]);
}

void trampoline()
{
    State& s = *g;
    const int me = s.cur;
    (*s.body)(me);
    // uniform control flow: the lanes finish one after the other without meeting again
    --s.live;
    void* dead;
    if (s.live == 0) { s.sp[me] = nullptr; simt_switch(&dead, s.mainSp); }
    int n = me;
    do { n = (n + 1) % WAVE; } while (n != me && (s.sp[n] == nullptr));
    s.sp[me] = nullptr; s.cur = n;
    simt_switch(&dead, s.sp[n]);
    abort();
}
}  // namespace

void run(const std::function<void(int)>& body)
{
    State st; memset(st.sp, 0, sizeof st.sp); memset(st.opCount, 0, sizeof st.opCount);
    st.stacks = (uint8_t*)mmap(nullptr, kStack * WAVE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (st.stacks == (uint8_t*)MAP_FAILED) { perror("simt: mmap"); abort(); }
    st.body = &body; st.live = WAVE;
    for (int l = 0; l < WAVE; ++l) {
        // initial frame: six callee-saved registers, then the return address; the stack is 16-byte aligned at the
        // trampoline's first instruction as after a call (rsp % 16 == 8)
        uint64_t* top = (uint64_t*)(st.stacks + kStack * (l + 1));
        top -= 1;                                   // alignment slot
        *--top = (uint64_t)(uintptr_t)&trampoline;  // ret target
        for (int k = 0; k < 6; ++k) *--top = 0;
        st.sp[l] = top;
    }
    State* prev = g; g = &st;
    st.cur = 0;
    simt_switch(&st.mainSp, st.sp[0]);
    g = prev;
    munmap(st.stacks, kStack * WAVE);
}

int lane() { return g->cur; }
void barrier() { next(); }

// every primitive: publish, meet, read.  The slots alternate between two sets so that a lane that runs ahead into the
// next primitive cannot overwrite a value a slower lane has still to read.
static inline uint64_t* publish(uint64_t v)
{
    State& s = *g;
    const int me = s.cur;
    uint64_t* set = s.slot[s.opCount[me]++ & 1u];
    set[me] = v;
    next();
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
