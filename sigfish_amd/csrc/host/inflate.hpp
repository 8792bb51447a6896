// inflate.hpp -- a fast zlib-stream (RFC 1950 / 1951) decoder for BLOW5 records.
//
// Why not zlib's inflate(): once the DTW runs on the GPU, decompressing the input is the largest item on the host
// (SURVEY.md 8f-1/8f-2).  zlib decodes a byte or a match at a time through a state machine that can stop anywhere;
// a BLOW5 record is a few KB that is always in memory as a whole, so this decoder keeps a 64-bit bit buffer refilled
// with unaligned 8-byte loads, resolves literal/length and distance codes through one 11-bit / 8-bit table lookup
// (second-level tables for longer codes), and copies matches eight bytes at a time.  Anything unusual -- preset
// dictionary, corrupt stream, Adler-32 mismatch -- makes it return false and the caller falls back to zlib.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sfa {

// Inflate the zlib stream [in, in+n) into *out (grown as needed, never shrunk); *len = bytes produced.
// Returns false on any malformed / unsupported input (the output is then unspecified).
bool fast_inflate_zlib(const uint8_t *in, size_t n, std::vector<uint8_t> *out, size_t *len);

// The same for TWO streams decoded side by side by one thread (their symbol loops take turns: two dependence chains keep a
// core busier than one); ok[k] false where stream k was declined, exactly as the single-stream call would.
void fast_inflate_zlib_pair(const uint8_t *const in[2], const size_t n[2], std::vector<uint8_t> *const out[2], size_t len[2], bool ok[2]);

}  // namespace sfa
