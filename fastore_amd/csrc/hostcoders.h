// Host-side (block 0 only) entry points of the coder cores; see hostcoders.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
namespace fshost {
void ppmdEncode(const uint8_t* in, size_t n, std::vector<uint8_t>& out);                        // one PPMd member
void rcEncode(uint32_t model, const uint8_t* pairs, size_t nPairs, std::vector<uint8_t>& out);  // one range-coded stream
void qvzEncode(const uint8_t* modelBlob, const uint8_t* symbols, size_t nSymbols, std::vector<uint8_t>& out);   // one QVZ quality stream
}
