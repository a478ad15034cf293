// Three-wave form of the PPMd encoder: the windows of the hit path are prepared AHEAD of the serial walk.
// Included by ppmd_core.h inside namespace fsppmd (64-lane builds only), after ppmd_window.h.
//
// Why.  A long quality stream is a chain of cycles: a window of ~40 plain hits (ppmd_window.h: ~9 000 clocks, most of
// them in fetch latencies and the rounds), then a serial episode of one to four symbols that the model has to be
// updated for (a read boundary: ~7 000 clocks of dependent record -> list -> suffix loads).  One wave does them one after
// the other; neither half can be made much shorter by itself (a wave issues one instruction per 4-5 clocks and every
// load is a 300-900 clock round trip), but they hardly touch the same memory: an episode rewrites the context it
// started in, that context's suffixes (lower orders: never in a window) and contexts it creates; a window rewrites the
// frequencies of full-order contexts.  So a wave of its own -- the WINDOW wave -- prepares the next window (fetch, chain,
// owners, rounds: everything but the stores) while the SERIAL wave walks the episode in front of it, from the memory as
// it was before the episode; when the serial wave arrives at the window's first position, the window wave checks that
// the guess was right and that the episode has not touched any of the window's contexts, and only then writes the lists
// back and hands the prices to the coder wave.  A window that fails the check is prepared again from the memory as it is
// now -- what the one- and two-wave forms do every time.  Bytes are those of the serial walk by construction: a prepared
// window is used only if it is the window the two-wave form would have computed at that point.
//
//   serial wave  S : encode_member() as ever, but window_step() becomes a request to W and a wait for its reply
//   coder wave   C : coder_wave(), unchanged; S and W take turns as its producers (never both at once)
//   window wave  W : window_wave() below
//
// Messages (LDS, one writer each, sequence-numbered; FS_Q_STORE / FS_Q_LOAD = release / acquire at workgroup scope):
//   request  S -> W   position, its context, UnitsStart, coder-ring tail, flags, the full-order contexts the last episode
//                     started in (it rewrites them) or created (they did not exist when W fetched)
//   forecast S -> W   where the running episode is expected to end (known after its first symbol's escapes: the order
//                     fell by d, each further symbol brings it back up by one) -- lets W choose the window's first lane
//   reply    W -> S   symbols coded, the context after them, PrevSuccess, the last four bytes, the new ring tail
// Global memory between the waves (same CU, shared vL1D): a wave's stores are complete (s_waitcnt vmcnt(0)) before it
// tells the other wave that they are there: W before its reply, S right after its request (off the critical path: it
// waits for the reply anyway) -- W does not fetch for the next window before that.
#pragma once

enum : uint32_t { WQ_EXIT = 1u, WQ_NEWSTREAM = 2u, WQ_RESTARTED = 4u, WQ_MAX_TOUCHED = 14u, WQ_MAX_START = 40u };
// request words
enum { WQ_POS = 0, WQ_CTX, WQ_HIST, WQ_UNITS, WQ_TAIL, WQ_FLAGS, WQ_N, WQ_IN_LO, WQ_IN_HI, WQ_HB_LO, WQ_HB_HI, WQ_SKIP /* full-order symbols coded without a request should this one find no window */,
       WQ_NT /* entries of wqT that count */, WQ_WORDS };      // (at most 16: one LDS read fetches them all, a lane each)
enum { WR_DONE = 0, WR_CTX, WR_PREV, WR_HIST, WR_TAIL, WR_WORDS };

#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
  #define FS_DRAIN_STORES() ((void)0)
#endif
// -DFS_SCOUT_PROFILE (with -DFS_WIN_PROFILE): the window wave's timeline in the phase slots -- [8] waiting for a request or for the
// serial wave's stores, [9] fetch, [10] solve, [11] waiting for the request behind a solved window, [12] check + prices + reply
// (+ the write-back when it must come first), [13] the write-back behind the reply, [14] the SERIAL wave waiting for replies
#if defined(FS_SCOUT_PROFILE)
  #define FS_SCOUT_PROF(w, t0) FS_PROF_ACC(w, t0)
#else
  #define FS_SCOUT_PROF(w, t0) ((void)(t0))
#endif
// test-only: extra meeting points in the lock-step emulation, to move the waves against each other (tests/test_simt.py)
#if defined(FS_SIMT_EMU)
  #define FS_EMU_JITTER(site) simt_jitter(site)
#else
  #define FS_EMU_JITTER(site) ((void)0)
#endif

// the serial wave and the window wave wait for each other on the step's critical path: short naps (s_sleep 1 = 64 clocks; the
// coder ring's s_sleep 8 would add up to 500 clocks to every hand-over), each wave on a SIMD of its own
#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_W_SPIN() __builtin_amdgcn_s_sleep(1)
#else
  #define FS_W_SPIN() ((void)0)
#endif

// ---- serial wave ----
// the serial path is about to code a symbol that starts at full order in context `c`: the episode touches it
FS_DEV void scout_touch(Coder& m, uint32_t c)
{
    const uint32_t t = FS_LDS_RD(m.sh->wq[WQ_NT]);
    if (FS_LANE() == 0) { if (t < WQ_MAX_TOUCHED) m.sh->wqT[t] = c; m.sh->wq[WQ_NT] = t + 1u; }
    FS_WAVE_SYNC();
}
// The window wave answers before its window's lists are back in memory when the context the serial walk goes on in is not
// one of them (window_wave: `late`).  The first symbol behind a reply needs nothing else of the window's; any later symbol
// that starts at full order may -- by then the stores are long complete, but it makes sure.
FS_DEV bool scout_lists_back(Coder& m) { return (int32_t)(FS_Q_LOAD(m.sh->wrDrained) - m.wSeq) >= 0; }
FS_DEV void scout_wait_lists_back(Coder& m)
{
    if (m.wHintDue != 0u) return;                                // the first symbol behind a reply: its context is not among the window's
    while (!scout_lists_back(m)) FS_W_SPIN();
}
// after the escapes of the first symbol behind a window: the order has fallen by OrderFall (each further symbol brings it back up
// by one), and `skip` further full-order symbols will be coded without asking for a window (encode_member: winSkip)
FS_DEV void scout_forecast(Coder& m, uint32_t pos, uint32_t skip)
{
    if (m.wHintDue == 0u) return;
    m.wHintDue = 0u;
    const uint32_t d = (uint32_t)m.OrderFall, e = pos - 1u + (d > 1u ? d : 1u) + skip;
    if (FS_LANE() == 0) m.sh->whPos = e;
    FS_WAVE_SYNC();
    FS_Q_STORE(m.sh->whSeq, m.wSeq + 1u);
}
FS_DEV void scout_post(Coder& m, uint32_t flags, fs_cgptr in, uint32_t n, uint32_t pos, uint32_t ctx, uint32_t hist, uint32_t skipIfNone)
{
    FS_LDS Shared* sh = m.sh;
    const uint32_t seq = m.wSeq + 1u;
    const uint32_t rs0 = sh->restarts, rs1 = sh->wsRestartsSeen;      // (both reads in flight together)
    const uint32_t rs = FS_LDS_RD(rs0);
    if (rs != FS_LDS_RD(rs1)) flags |= WQ_RESTARTED;
    if (FS_LANE() == 0) {
        const uint64_t i = (uint64_t)(uintptr_t)in, h = (uint64_t)(uintptr_t)m.hb;
        sh->wq[WQ_POS] = pos; sh->wq[WQ_CTX] = ctx; sh->wq[WQ_HIST] = hist; sh->wq[WQ_UNITS] = m.UnitsStart; sh->wq[WQ_TAIL] = m.qTail; sh->wq[WQ_FLAGS] = flags;
        sh->wq[WQ_N] = n; sh->wq[WQ_IN_LO] = (uint32_t)i; sh->wq[WQ_IN_HI] = (uint32_t)(i >> 32); sh->wq[WQ_HB_LO] = (uint32_t)h; sh->wq[WQ_HB_HI] = (uint32_t)(h >> 32); sh->wq[WQ_SKIP] = skipIfNone;
        sh->wsRestartsSeen = rs;
    }
    FS_WAVE_SYNC();
    FS_Q_STORE(sh->wqSeq, seq);
    m.wSeq = seq;
    // this wave's stores (the episode's model updates) are complete before the window wave fetches again
    FS_DRAIN_STORES();
    FS_Q_STORE(sh->wsDrained, seq);
}
// window_step() of the three-wave form: ask the window wave, wait for its answer
FS_DEV uint32_t scout_window(Coder& m, fs_cgptr in, uint32_t n, uint32_t pos, uint32_t MinContext, uint32_t& hist, uint32_t skipIfNone)
{
    FS_LDS Shared* sh = m.sh;
    const uint32_t fresh = FS_LDS_RD(sh->wsNewStream);
    scout_post(m, fresh ? (uint32_t)WQ_NEWSTREAM : 0u, in, n, pos, MinContext, hist, skipIfNone);
    const uint32_t seq = m.wSeq;
    FS_EMU_JITTER(0);
    uint64_t tw = FS_PROF_NOW();
    while (FS_Q_LOAD(sh->wrSeq) != seq) FS_W_SPIN();
    FS_SCOUT_PROF(sh->winStats[14], tw);
    // the reply's words with one LDS read, a lane each
    const uint32_t rw = sh->wr[(uint32_t)FS_LANE() & 7u];
    FS_EMU_MEET();
    const uint32_t done = FS_UNI(fs_readlane(rw, WR_DONE));
    if (done != 0u) { m.MaxContext = FS_UNI(fs_readlane(rw, WR_CTX)); m.PrevSuccess = FS_UNI(fs_readlane(rw, WR_PREV)); hist = FS_UNI(fs_readlane(rw, WR_HIST)); }
    m.qTail = FS_UNI(fs_readlane(rw, WR_TAIL));
    if (FS_LANE() == 0) { sh->wq[WQ_NT] = 0u; sh->wsNewStream = 0u; }
    m.wHintDue = 1u;
    FS_WAVE_SYNC();
    return done;
}
// a new stream begins (encode_member): its first request tells the window wave
FS_DEV void scout_begin_stream(Coder& m)
{
    if (FS_LANE() == 0) { m.sh->wsNewStream = 1u; m.sh->wq[WQ_NT] = 0u; m.sh->wsRestartsSeen = 0u; for (int i = 0; i < 4; ++i) m.sh->wxStats[i] = 0u; }      // (the window wave may be adding to them: statistics only)
    m.wHintDue = 0u; m.wSeq = FS_LDS_RD(m.sh->wqSeq);
    FS_WAVE_SYNC();
}
// the serial wave's last word to the window wave
FS_DEV void scout_send_exit(FS_LDS Shared* sh)
{
    Coder m; m.sh = sh; m.hb = nullptr; m.UnitsStart = 0; m.qTail = 0; m.wSeq = FS_LDS_RD(sh->wqSeq);
    scout_post(m, WQ_EXIT, nullptr, 0u, 0u, 0u, 0u, 0u);
}
// before the first request of a workgroup
FS_DEV void scout_init(FS_LDS Shared* sh)
{
    sh->wqSeq = 0u; sh->whSeq = 0u; sh->whPos = 0u; sh->wsDrained = 0u; sh->wrSeq = 0u; sh->wrDrained = 0u; sh->wq[WQ_NT] = 0u; sh->wsNewStream = 1u; sh->wsRestartsSeen = 0u;
    for (int i = 0; i < 4; ++i) sh->wxStats[i] = 0u;
}

// ---- window wave ----
// win_solve's look at the mailboxes (ppmd_window.h): nonzero when forecast or request number `watchSeq` is there and names a
// position other than `at`
FS_DEV uint32_t win_watch(Coder& m, uint32_t watchSeq, uint32_t at)
{
    if (FS_Q_LOAD(m.sh->wqSeq) == watchSeq) return FS_LDS_RD(m.sh->wq[WQ_POS]) != at ? 1u : 0u;
    if (FS_Q_LOAD(m.sh->whSeq) == watchSeq) return FS_LDS_RD(m.sh->whPos) != at ? 1u : 0u;
    return 0u;
}
#if defined(FS_SIMT_EMU)
  #define FS_SCOUT_HIST(actual, guess) do { if (FS_LANE() == 0) simt_scout_hist(actual, guess); } while (0)
  #define FS_SCOUT_WHY(i, c) do { if ((c) && FS_LANE() == 0) simt_scout_why(i); } while (0)
#else
  #define FS_SCOUT_WHY(i, c) ((void)0)
  #define FS_SCOUT_HIST(actual, guess) ((void)0)
#endif
FS_DEV void window_wave(FS_LDS Shared* sh)
{
    const uint32_t lane = (uint32_t)FS_LANE();
    Coder m;
    m.sh = sh; m.hb = nullptr; m.queued = 1u; m.qTail = 0u; m.qHeadSeen = 0u; m.inAhead = 0u; m.UnitsStart = 0u;
    fs_cgptr in = nullptr; uint32_t n = 0;
    uint32_t seen = 0;                       // requests read
    bool fresh = true, haveReq = false;     // fresh: the next window starts at a request's position, in its context
    bool guessed = false;                   // the start lane of the window being solved is this wave's own guess
    bool usedAhead = false;
    uint32_t base = 0, s = 0;
    uint32_t rPos = 0, rCtx = 0, rHist = 0, rUnits = 0, rTail = 0, rFlags = 0, rNT = 0, rSkip = 0, guess = 0;
    uint32_t rT = 0;                          // lane t (< 16): the t-th context of the request's list
    // (all of a request with two LDS reads, a lane per word)
    #define FS_READ_REQUEST() do { \
        const uint32_t rq_ = sh->wq[lane & 15u]; rT = sh->wqT[lane & 15u]; FS_EMU_MEET(); \
        rPos = FS_UNI(fs_readlane(rq_, WQ_POS)); rCtx = FS_UNI(fs_readlane(rq_, WQ_CTX)); rHist = FS_UNI(fs_readlane(rq_, WQ_HIST)); rUnits = FS_UNI(fs_readlane(rq_, WQ_UNITS)); \
        rTail = FS_UNI(fs_readlane(rq_, WQ_TAIL)); rFlags = FS_UNI(fs_readlane(rq_, WQ_FLAGS)); rNT = FS_UNI(fs_readlane(rq_, WQ_NT)); rSkip = FS_UNI(fs_readlane(rq_, WQ_SKIP)); \
        rN = FS_UNI(fs_readlane(rq_, WQ_N)); rInLo = FS_UNI(fs_readlane(rq_, WQ_IN_LO)); rInHi = FS_UNI(fs_readlane(rq_, WQ_IN_HI)); rHbLo = FS_UNI(fs_readlane(rq_, WQ_HB_LO)); rHbHi = FS_UNI(fs_readlane(rq_, WQ_HB_HI)); \
        ++seen; haveReq = true; } while (0)
    uint32_t rN = 0, rInLo = 0, rInHi = 0, rHbLo = 0, rHbHi = 0;
    uint64_t tq = FS_PROF_NOW();
    for (;;) {
        if (fresh && !haveReq) {
            while (FS_Q_LOAD(sh->wqSeq) != seen + 1u) FS_W_SPIN();
            FS_READ_REQUEST();
        }
        if (haveReq && (rFlags & WQ_EXIT) != 0u) return;
        if (haveReq && (rFlags & WQ_NEWSTREAM) != 0u) {
            const uint64_t i = (uint64_t)rInLo | ((uint64_t)rInHi << 32), h = (uint64_t)rHbLo | ((uint64_t)rHbHi << 32);
            in = (fs_cgptr)(uintptr_t)i; m.hb = (fs_gptr)(uintptr_t)h; n = rN; m.inAhead = 0u;
            fresh = true;
        }
        if (fresh) { base = rPos; s = 0u; m.UnitsStart = rUnits; m.qTail = rTail; }
        else if (base + 8u >= n) { fresh = true; continue; }           // nothing worth looking ahead at: wait for the serial wave
        // the serial wave's stores up to its last request are complete
        while ((int32_t)(FS_Q_LOAD(sh->wsDrained) - seen) < 0) FS_W_SPIN();
        FS_EMU_JITTER(1);
        FS_SCOUT_PROF(sh->winStats[8], tq);
        uint64_t tp = FS_PROF_NOW();
        WinFetch<8> f; WinSolved<8> o;
        win_fetch(m, in, n, base, fresh ? rCtx : 0u, f);
        FS_STAT_ADD(sh->winStats[0], 1u);
        FS_PROF_ACC_W(sh->winStats[8], tp);
        FS_SCOUT_PROF(sh->winStats[9], tq);
        if (!fresh) {
            // the window's first lane: where the serial wave IS, if it is there already; else where it says it will be back at full
            // order, if it has said so; else this wave's own guess -- most episodes are one symbol long (the order falls by one
            // level at most), and after an attempt without a window the serial wave codes `skip` more symbols unasked.  Waiting
            // for the forecast would cost more (it comes a third into the episode) than the guess loses: a window that starts
            // too early is solved again when the request is there.
            uint32_t e;
            guessed = false;
            if (FS_Q_LOAD(sh->wqSeq) == seen + 1u) { FS_READ_REQUEST(); e = rPos; }
            else if (FS_Q_LOAD(sh->whSeq) == seen + 1u) e = FS_LDS_RD(sh->whPos);
#if defined(FS_SCOUT_WAIT_FORECAST)
            else {      // (experiment: no guess -- wait for the forecast or the request)
                for (;;) {
                    if (FS_Q_LOAD(sh->wqSeq) == seen + 1u) { FS_READ_REQUEST(); e = rPos; break; }
                    if (FS_Q_LOAD(sh->whSeq) == seen + 1u) { e = FS_LDS_RD(sh->whPos); break; }
                    FS_W_SPIN();
                }
            }
#else
            else { e = base + guess; guessed = true; }
#endif
            if ((haveReq && (rFlags & (WQ_EXIT | WQ_NEWSTREAM | WQ_RESTARTED)) != 0u) || e < base || e - base > (uint32_t)WQ_MAX_START) { FS_STAT_ADD(sh->wxStats[1], 1u); fresh = true; continue; }
            s = e - base;
        }
        FS_EMU_JITTER(2);
        uint32_t E;
        bool valid = true;
        for (uint32_t pass = 0;; ++pass) {
            E = win_solve<8>(m, in, n, base, f, s, o, tp, (!fresh && guessed && !haveReq) ? seen + 1u : 0u);
            FS_SCOUT_PROF(sh->winStats[10], tq);
            if (fresh) break;
            if (E == (uint32_t)WIN_ABORT) {
                // the serial wave has said (forecast) or shown (request) meanwhile that it comes back elsewhere: solved again from there
                uint32_t e2;
                if (FS_Q_LOAD(sh->wqSeq) == seen + 1u) { FS_READ_REQUEST(); e2 = rPos; } else e2 = FS_LDS_RD(sh->whPos);
                guessed = false;
                FS_STAT_ADD(sh->wxStats[3], 1u);
                if ((haveReq && (rFlags & (WQ_EXIT | WQ_NEWSTREAM | WQ_RESTARTED)) != 0u) || e2 < base || e2 - base > (uint32_t)WQ_MAX_START || e2 - base >= f.W) { valid = false; E = s; break; }
                s = e2 - base; pass = 0u - 1u;
                continue;
            }
            if (!haveReq) {
                while (FS_Q_LOAD(sh->wqSeq) != seen + 1u) FS_W_SPIN();
                FS_READ_REQUEST();
            }
            FS_SCOUT_PROF(sh->winStats[11], tq);
            valid = (rFlags & (WQ_EXIT | WQ_NEWSTREAM | WQ_RESTARTED)) == 0u && rNT <= (uint32_t)WQ_MAX_TOUCHED;
            FS_SCOUT_WHY(0, (rFlags & (WQ_EXIT | WQ_NEWSTREAM | WQ_RESTARTED)) != 0u); FS_SCOUT_WHY(3, rNT > (uint32_t)WQ_MAX_TOUCHED);
            if (!valid) break;
            FS_SCOUT_HIST(pass == 0u ? (rPos >= base ? rPos - base : 63u) : 64u, s);
            if (rPos != base + s) {
                // the episode took longer than forecast: what was fetched for the later lanes is as good as before -- the window
                // is solved again from the lane the serial wave really stands at (once)
                FS_SCOUT_WHY(1, true); FS_SCOUT_WHY(7, rPos < base + s);
                if (pass == 0u && rPos > base + s && rPos - base <= (uint32_t)WQ_MAX_START && rPos - base < f.W) { s = rPos - base; FS_STAT_ADD(sh->wxStats[3], 1u); continue; }
                valid = false; break;
            }
            // the serial wave stands at lane s: in the context fetched for it, and none of the window's contexts rewritten since?
            valid = FS_UNI(fs_readlane(f.addr, s)) == rCtx;
            FS_SCOUT_WHY(4, !valid);
            if (!valid) break;
            {
                const uint32_t last = E > s ? E : s + 1u;            // (an empty window: lane s alone decides)
                bool hit = false;
                for (uint32_t t = 0; t < rNT; ++t) { const uint32_t c = FS_UNI(fs_readlane(rT, t)); hit = hit || (lane >= s && lane < last && f.addr == c); }
                valid = fs_ballot(hit) == 0ull; FS_SCOUT_WHY(5, !valid);
            }
            FS_SCOUT_WHY(2, valid && E <= s);
            // (valid and E <= s: the position is no plain hit in its true context, fetched before the episode and untouched by
            // it -- a fresh look would find the same: the answer is "none" right away)
            break;
        }
        if (!fresh) {
            if (!valid) { FS_STAT_ADD(sh->wxStats[2], 1u); fresh = true; continue; }
            usedAhead = true;
            m.UnitsStart = rUnits; m.qTail = rTail;
        }
        const uint32_t done = E > s ? E - s : 0u;
        uint32_t mc = 0u, ps = 0u, hh = rHist;
        bool late = false;                                       // the lists go back AFTER the reply
        if (done != 0u) {
            // the prices of lanes [s, E) in stream order: in the coder's ring before the serial wave adds its own
            const uint32_t tA = s ? fs_bperm(o.tA, (lane + s) & 63u) : o.tA, tM = s ? fs_bperm(o.tM, (lane + s) & 63u) : o.tM;
            cq_push_lanes(m, tA, tM, done);
            ps = (FS_UNI(fs_readlane(o.tA, E - 1u)) >> 23) & 1u;
            mc = FS_UNI(fs_readlane(f.succ, E - 1u));
            const uint32_t kl = FS_UNI(fs_readlane(f.key, E - 1u)), sl = FS_UNI(fs_readlane(f.sym, E - 1u));
            hh = (kl >> 8) | (sl << 24);
            // The serial wave goes on in context `mc`, its suffixes (lower orders: no window ever writes them) and contexts
            // it makes.  Unless `mc` is one of THIS window's contexts, it need not wait for the lists to be back: the reply
            // goes first, the stores follow while it walks (before it looks at any other full-order context it waits for
            // wrDrained: scout_touch).
            late = fs_ballot(lane >= s && lane < E && f.addr == mc) == 0ull;
            if (!late) {
                win_write_back<8>(m, f, o);
                FS_DRAIN_STORES();                               // the lists are where the serial wave will read them
            }
            FS_PROF_ACC_W(sh->winStats[12], tp);
        }
        if (lane == 0u) { sh->wr[WR_DONE] = done; sh->wr[WR_CTX] = mc; sh->wr[WR_PREV] = ps; sh->wr[WR_HIST] = hh; sh->wr[WR_TAIL] = m.qTail; }
        FS_WAVE_SYNC();
        if (!late) FS_Q_STORE(sh->wrDrained, seen);
        FS_Q_STORE(sh->wrSeq, seen);
        FS_SCOUT_PROF(sh->winStats[12], tq);
        if (done != 0u) { FS_STAT_ADD(sh->winStats[1], 1u); FS_STAT_ADD(sh->winStats[2], done); }      // (counters: behind the reply, off the serial wave's wait)
        if (usedAhead) { FS_STAT_ADD(sh->wxStats[0], 1u); usedAhead = false; }
        if (late) {
            FS_EMU_JITTER(3);
            win_write_back<8>(m, f, o);
            FS_DRAIN_STORES();
            FS_Q_STORE(sh->wrDrained, seen);
            FS_SCOUT_PROF(sh->winStats[13], tq);
        }
        // look ahead: the serial wave now codes the symbol at rPos + done; the positions behind it are fetched meanwhile
        haveReq = false; fresh = false; base = rPos + done + 1u; guess = done != 0u ? 0u : rSkip;
    }
    #undef FS_READ_REQUEST
}
