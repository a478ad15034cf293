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
    uint64_t position() const { return pos_; }
    uint64_t size() const { return size_; }
    const uint8_t* data() const { return p_; }
    uint32_t getByte() { if (pos_ >= size_) throw std::runtime_error("bin stream truncated"); return p_[pos_++]; }
    uint32_t getBit()
    {
        if (wpos_ == 0) { word_ = getByte(); wpos_ = 7; return (word_ >> 7) & 1; }
        return (word_ >> (--wpos_)) & 1;
    }
    uint32_t get2Bits()
    {
        if (wpos_ >= 2) { wpos_ -= 2; return (word_ >> wpos_) & 3; }
        if (wpos_ == 0) { word_ = getByte(); wpos_ = 6; return (word_ >> wpos_) & 3; }
        uint32_t w = (word_ & 1) << 1;
        word_ = getByte(); wpos_ = 7;
        w += word_ >> wpos_;
        return w & 3;
    }
    uint32_t getBits(uint32_t n)
    {
        uint32_t w = 0;
        while (n) {
            if (wpos_ == 0) { word_ = getByte(); wpos_ = 8; }
            if (n > wpos_) { w <<= wpos_; w += word_ & ((1u << wpos_) - 1); n -= wpos_; wpos_ = 0; }
            else { w <<= n; wpos_ -= n; w += (word_ >> wpos_) & ((1u << n) - 1); break; }
        }
        return w;
    }
    void getBytes(void* dst, uint64_t n) { if (pos_ + n > size_) throw std::runtime_error("bin stream truncated"); memcpy(dst, p_ + pos_, n); pos_ += n; }
    uint32_t get2Bytes() { uint32_t a = getByte(); return (a << 8) | getByte(); }
    uint32_t get4Bytes() { uint32_t c = getByte(); c = (c << 8) | getByte(); c = (c << 8) | getByte(); return (c << 8) | getByte(); }
    uint64_t get8Bytes() { uint64_t c = 0; for (int i = 0; i < 8; ++i) c = (c << 8) | getByte(); return c; }
    void flushWord() { wpos_ = 0; }
private:
    const uint8_t* p_; uint64_t size_; uint64_t pos_ = 0; uint32_t word_ = 0, wpos_ = 0;
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
