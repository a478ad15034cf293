// TEST-ONLY: the emission kernels' body (fastore_amd/csrc/emit_wave.h: a wavefront per op, match bits as packed ballots) on the
// lock-step emulation of tests/emu/simt.h.  Part of build/libsimt_emu.so; the emulation library's stand-in for fs_emit_* calls it
// when FS_EMU_SIMT_EMIT names that library, so that the product's own parity check (fsgpu_emit_check: pre-entropy bytes of every
// stream against the host walk's) and whole archives run over the 64-lane code without a GPU.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "simt.h"
#include "../../fastore_amd/csrc/emit_wave.h"

// One bin: count pass, the per-channel sums (what fs_emit_scan does with block scans), write pass.  Letters and match symbols land
// where the emulation keeps them (out + out_off[c]); the packed match bits are unpacked into the emulation's byte-per-bit scratch
// (out + raw_off[c]), whose run-length coder is the serial one.  at[c]: what every channel holds afterwards.
extern "C" int simt_emit_job(const uint8_t* buf, const fsdev::EmitJob* jobp, const fsdev::EmitOp* ops, uint8_t* out, uint32_t* at)
{
    using namespace fsdev;
    const EmitJob& job = *jobp;
    const uint32_t n = job.n_ops;
    std::vector<uint32_t> cL(n), cB(n), oL(n), oB(n);
    simt::run([&](int lane) {
        for (uint32_t k = 0; k < n; ++k) {
            const fsemit::WaveCount c = fsemit::emit_op_wave<false>(ops[job.first_op + k], job, buf + job.seq_off, buf + job.contig_off, nullptr, nullptr, nullptr, 0u);
            if (lane == 0) { cL[k] = c.nL; cB[k] = c.nB; }
        }
    });
    for (uint32_t c = 0; c < ECH_COUNT; ++c) at[c] = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const EmitOp& op = ops[job.first_op + k];
        const uint32_t chL = fsemit::channel_l(op), chB = fsemit::channel_b(op);
        if (chL < ECH_COUNT) { oL[k] = at[chL]; at[chL] += cL[k]; } else if (cL[k]) return -1;
        if (chB < ECH_COUNT) { oB[k] = at[chB]; at[chB] += cB[k]; } else if (cB[k]) return -1;
    }
    std::vector<std::vector<uint32_t>> words(ECH_COUNT);
    for (uint32_t c = 0; c < ECH_COUNT; ++c) if (fsemit::is_bit_channel(c)) words[c].assign((at[c] + 31u) / 32u + 2u, 0u);
    simt::run([&](int) {
        for (uint32_t k = 0; k < n; ++k) {
            const EmitOp& op = ops[job.first_op + k];
            const uint32_t chL = fsemit::channel_l(op), chB = fsemit::channel_b(op);
            uint8_t* outL = chL < ECH_COUNT ? out + job.out_off[chL] + (uint64_t)fsemit::unit_l(chL) * oL[k] : nullptr;
            const bool bitCh = chB < ECH_COUNT && fsemit::is_bit_channel(chB);
            uint8_t* outSym = (chB < ECH_COUNT && !bitCh) ? out + job.out_off[chB] + 2ull * oB[k] : nullptr;
            uint32_t* outBits = bitCh ? words[chB].data() : nullptr;
            (void)fsemit::emit_op_wave<true>(op, job, buf + job.seq_off, buf + job.contig_off, outL, outSym, outBits, oB[k]);
            simt::barrier();                               // (ops of one channel share boundary words: one op's ORs are done before the next op's)
        }
    });
    for (uint32_t c = 0; c < ECH_COUNT; ++c)
        if (fsemit::is_bit_channel(c) && job.item[c] != 0xFFFFFFFFu)
            for (uint32_t i = 0; i < at[c]; ++i) out[job.raw_off[c] + i] = (uint8_t)((words[c][i >> 5] >> (i & 31u)) & 1u);
    return 0;
}
