// MSB-first bit reader and big-endian byte writer matching the reference's BitMemoryReader /
// BitMemoryWriter byte-level behaviour (/root/reference/fastore/fastore_bin/BitMemory.h:23-196,
// 203-436).
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdexcept>
#include <vector>

namespace fs {

class BitReader {
public:
    BitReader(const uint8_t* p, uint64_t size) : p_(p), size_(size) {}
    // bytes consumed, counting a partly read byte as consumed (the reference's reader holds it in its word buffer)
    uint64_t position() const { return pos_ - cnt_ / 8; }
    uint64_t size() const { return size_; }
    const uint8_t* data() const { return p_; }
    uint32_t getBit() { return getBits(1); }
    uint32_t get2Bits() { return getBits(2); }
    // n <= 32.  The window holds up to 64 bits, MSB first, refilled a byte at a time, so that dropping the rest of
    // a partly read byte (flushWord) is just dropping cnt_ % 8 bits.
    uint32_t getBits(uint32_t n)
    {
        if (n == 0) return 0;
        if (cnt_ < n) refill(n);
        const uint32_t v = (uint32_t)(acc_ >> (64 - n));
        acc_ <<= n; cnt_ -= n;
        return v;
    }
    uint32_t getByte() { return getBits(8); }
    void getBytes(void* dst, uint64_t n)
    {
        uint8_t* d = (uint8_t*)dst;
        flushWord();                                                 // the rest of a partly read byte is not part of a byte run
        while (n && cnt_ >= 8) { *d++ = (uint8_t)(acc_ >> 56); acc_ <<= 8; cnt_ -= 8; --n; }      // whole bytes still in the window
        if (n == 0) return;
        cnt_ = 0; acc_ = 0;                                          // callers read byte runs at byte boundaries
        if (pos_ + n > size_) throw std::runtime_error("bin stream truncated");
        memcpy(d, p_ + pos_, n); pos_ += n;
    }
    uint32_t get2Bytes() { return getBits(16); }
    uint32_t get4Bytes() { return getBits(32); }
    uint64_t get8Bytes() { const uint64_t hi = getBits(32); return (hi << 32) | getBits(32); }
    void flushWord() { const uint32_t r = cnt_ & 7u; acc_ <<= r; cnt_ -= r; }
    // bits consumed so far, and a skip over n bits that are read elsewhere (the device-side quality path)
    uint64_t bitPosition() const { return pos_ * 8u - cnt_; }
    void skipBits(uint64_t n)
    {
        if (n <= cnt_) { if (n >= 64) { acc_ = 0; cnt_ = 0; } else { acc_ <<= n; cnt_ -= (uint32_t)n; } return; }
        n -= cnt_; cnt_ = 0; acc_ = 0;
        if (n > (size_ - pos_) * 8u) throw std::runtime_error("bin stream truncated");
        pos_ += n >> 3;
        if (n & 7u) (void)getBits((uint32_t)(n & 7u));
    }
    // n 6-bit fields -> bytes (+ add): eight fields per 48-bit bite of the window
    void unpack6(uint8_t* dst, uint32_t n, uint32_t add)
    {
        while (n >= 8) {
            if (cnt_ < 48) {
                if (pos_ + 8 > size_) break;
                uint64_t w; memcpy(&w, p_ + pos_, 8); w = __builtin_bswap64(w);
                const uint32_t take = (64u - cnt_) >> 3;                   // whole bytes that fit behind the valid bits
                acc_ |= cnt_ ? (w >> cnt_) : w;
                pos_ += take; cnt_ += 8u * take;
                if (cnt_ < 64) acc_ &= ~0ull << (64 - cnt_);                // drop the partial byte that came along
            }
            const uint64_t x = acc_ >> 16;
            dst[0] = (uint8_t)(((x >> 42) & 63) + add); dst[1] = (uint8_t)(((x >> 36) & 63) + add); dst[2] = (uint8_t)(((x >> 30) & 63) + add); dst[3] = (uint8_t)(((x >> 24) & 63) + add);
            dst[4] = (uint8_t)(((x >> 18) & 63) + add); dst[5] = (uint8_t)(((x >> 12) & 63) + add); dst[6] = (uint8_t)(((x >> 6) & 63) + add); dst[7] = (uint8_t)((x & 63) + add);
            acc_ <<= 48; cnt_ -= 48; dst += 8; n -= 8;
        }
        for (; n; --n) *dst++ = (uint8_t)(getBits(6) + add);
    }
private:
    void refill(uint32_t need)
    {
        if (pos_ + 8 <= size_ && cnt_ <= 32) {                       // fast path: four bytes at once
            const uint64_t w = ((uint64_t)p_[pos_] << 24) | ((uint64_t)p_[pos_ + 1] << 16) | ((uint64_t)p_[pos_ + 2] << 8) | (uint64_t)p_[pos_ + 3];
            acc_ |= w << (32 - cnt_); cnt_ += 32; pos_ += 4;
            return;
        }
        while (cnt_ <= 56 && pos_ < size_) { acc_ |= (uint64_t)p_[pos_++] << (56 - cnt_); cnt_ += 8; }
        if (cnt_ < need) throw std::runtime_error("bin stream truncated");
    }
    const uint8_t* p_; uint64_t size_; uint64_t pos_ = 0; uint64_t acc_ = 0; uint32_t cnt_ = 0;
};

struct ByteWriter {
    std::vector<uint8_t> b;
    void put(uint32_t v) { b.push_back((uint8_t)v); }
    void put2(uint32_t v) { put(v >> 8); put(v & 0xFF); }
    void put4(uint32_t v) { put(v >> 24); put((v >> 16) & 0xFF); put((v >> 8) & 0xFF); put(v & 0xFF); }
    void put8(uint64_t v) { for (int i = 7; i >= 0; --i) put((uint32_t)(v >> (8 * i)) & 0xFF); }
    void putBytes(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
    size_t size() const { return b.size(); }
};

}  // namespace fs
