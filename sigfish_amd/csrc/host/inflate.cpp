// inflate.cpp -- see inflate.hpp.  Written against RFC 1950 (zlib wrapper) and RFC 1951 (DEFLATE) only.
#include "inflate.hpp"

#include <immintrin.h>

#include <cstring>

namespace sfa {
namespace {

// Adler-32 (RFC 1950) of the inflated record, 32 bytes per step with SSSE3 (zlib's adler32() on this image is the scalar
// loop: 1.9 us per 6 KB record, a tenth of the whole decode); scalar tail and fallback for machines without SSSE3.
inline uint32_t adler32_scalar(uint32_t adler, const uint8_t *p, size_t n) {
    uint32_t a = adler & 0xffffu, b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;  // the largest run for which b cannot overflow 32 bits
        n -= k;
        while (k--) {
            a += *p++;
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}
__attribute__((target("ssse3"))) uint32_t adler32_ssse3(uint32_t adler, const uint8_t *p, size_t n) {
    uint32_t a = adler & 0xffffu, b = adler >> 16;
    const __m128i w_hi = _mm_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17);
    const __m128i w_lo = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m128i ones = _mm_set1_epi16(1), zero = _mm_setzero_si128();
    while (n >= 32) {
        size_t blocks = n / 32;
        if (blocks > 5552 / 32) blocks = 5552 / 32;
        n -= blocks * 32;
        // over a run of 32-byte blocks: a' = a + sum(bytes); b' = b + 32 * (a at the start of every block, summed) + sum of the weighted bytes
        __m128i v_a = zero, v_b = zero, v_a_sum = zero;  // byte sums so far (per lane), weighted sums, sum over blocks of "byte sums before the block"
        for (size_t i = 0; i < blocks; ++i, p += 32) {
            const __m128i x0 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p)), x1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16));
            v_a_sum = _mm_add_epi32(v_a_sum, v_a);
            v_a = _mm_add_epi32(v_a, _mm_add_epi32(_mm_sad_epu8(x0, zero), _mm_sad_epu8(x1, zero)));
            v_b = _mm_add_epi32(v_b, _mm_madd_epi16(_mm_maddubs_epi16(x0, w_hi), ones));
            v_b = _mm_add_epi32(v_b, _mm_madd_epi16(_mm_maddubs_epi16(x1, w_lo), ones));
        }
        auto hsum = [](__m128i v) {
            v = _mm_add_epi32(v, _mm_shuffle_epi32(v, 0x4e));
            v = _mm_add_epi32(v, _mm_shuffle_epi32(v, 0xb1));
            return static_cast<uint32_t>(_mm_cvtsi128_si32(v));
        };
        b += static_cast<uint32_t>(blocks) * 32u * a + 32u * hsum(v_a_sum) + hsum(v_b);
        a += hsum(v_a);
        a %= 65521u;
        b %= 65521u;
    }
    return n ? adler32_scalar((b << 16) | a, p, n) : ((b << 16) | a);
}
inline uint32_t adler32_fast(const uint8_t *p, size_t n) {
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3");
    return have_ssse3 ? adler32_ssse3(1u, p, n) : adler32_scalar(1u, p, n);
}

constexpr int kLitBits = 10;   // primary lookup width of the literal/length table
constexpr int kDistBits = 8;   // ... of the distance table
constexpr int kClenBits = 7;   // ... of the code-length alphabet (codes are at most 7 bits)
constexpr int kMaxCodeLen = 15;

enum Kind : uint32_t { kLiteral = 0, kLength = 1, kEndOfBlock = 2, kSubtable = 3, kInvalid = 4, kDistance = 5 };

// entry: bits 0-3 code bits to consume, 4-7 extra bits (or subtable index bits), 8-10 kind, 16-31 value
inline uint32_t make_entry(uint32_t bits, uint32_t extra, uint32_t kind, uint32_t value) {
    return bits | (extra << 4) | (kind << 8) | (value << 16);
}
inline uint32_t e_bits(uint32_t e) { return e & 15u; }
inline uint32_t e_extra(uint32_t e) { return (e >> 4) & 15u; }
inline uint32_t e_kind(uint32_t e) { return (e >> 8) & 7u; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) {
        r = (r << 1) | (code & 1u);
        code >>= 1;
    }
    return r;
}

// what a decoded symbol means, per alphabet
inline uint32_t symbol_entry(int alphabet, int sym, uint32_t bits) {
    if (alphabet == 0) {  // literal / length
        if (sym < 256) return make_entry(bits, 0, kLiteral, static_cast<uint32_t>(sym));
        if (sym == 256) return make_entry(bits, 0, kEndOfBlock, 0);
        if (sym <= 285) return make_entry(bits, kLenExtra[sym - 257], kLength, kLenBase[sym - 257]);
        return make_entry(bits, 0, kInvalid, 0);
    }
    if (alphabet == 1) {  // distance
        if (sym < 30) return make_entry(bits, kDistExtra[sym], kDistance, kDistBase[sym]);
        return make_entry(bits, 0, kInvalid, 0);
    }
    return make_entry(bits, 0, kLiteral, static_cast<uint32_t>(sym));  // code-length alphabet: the value is the symbol
}

// Canonical Huffman decode table: `primary` bits resolved directly, longer codes through subtables appended behind
// the 2^primary first entries.  Unused slots stay kInvalid.  Returns false on an over-subscribed code.
// Symbols are visited in code order (by length, then by value); the bit-reversed code -- the table index, DEFLATE packs
// codes starting from their most significant bit -- is advanced with a reversed increment instead of being recomputed.
bool build_table(const uint8_t *lens, int n_sym, int alphabet, int primary, uint32_t *table, int capacity) {
    int count[kMaxCodeLen + 1] = {0};
    for (int s = 0; s < n_sym; ++s) count[lens[s]]++;
    count[0] = 0;
    int left = 1, max_len = 0;
    for (int l = 1; l <= kMaxCodeLen; ++l) {  // Kraft: never more codes of a length than remain
        left = (left << 1) - count[l];
        if (left < 0) return false;
        if (count[l]) max_len = l;
    }
    uint16_t offs[kMaxCodeLen + 2], sorted[288];
    offs[1] = 0;
    for (int l = 1; l <= kMaxCodeLen; ++l) offs[l + 1] = static_cast<uint16_t>(offs[l] + count[l]);
    const int n_codes = offs[kMaxCodeLen + 1];
    for (int s = 0; s < n_sym; ++s)
        if (lens[s]) sorted[offs[lens[s]]++] = static_cast<uint16_t>(s);
    const uint32_t psize = 1u << primary;
    const uint32_t invalid = make_entry(0, 0, kInvalid, 0);
    if (left > 0 || max_len < primary)  // incomplete code (or nothing at all): some slots stay unassigned
        for (uint32_t i = 0; i < psize; ++i) table[i] = invalid;
    uint32_t rev = 0;  // bit-reversed code of the current symbol, `len` bits
    int idx = 0;
    int len = 1;
    while (len <= kMaxCodeLen && !count[len]) ++len;
    uint32_t used = psize;
    uint32_t cur_prefix = ~0u, sub_base = 0, sub_bits = 0;
    int remaining = n_codes ? count[len] : 0;
    for (; idx < n_codes; ++idx) {
        const int sym = sorted[idx];
        if (len <= primary) {
            const uint32_t e = symbol_entry(alphabet, sym, static_cast<uint32_t>(len));
            for (uint32_t i = rev; i < psize; i += 1u << len) table[i] = e;
        } else {
            const uint32_t prefix = rev & (psize - 1);
            if (prefix != cur_prefix) {  // codes come in increasing order: all codes sharing a prefix are consecutive
                // subtable width: enough for the longest code under this prefix = grow while the codes of the current
                // and following lengths do not fill it (same rule as a canonical decoder's table sizing)
                cur_prefix = prefix;
                sub_bits = static_cast<uint32_t>(len - primary);
                int room = 1 << sub_bits, l2 = len;
                int cnt = remaining;
                while (l2 < max_len) {
                    room -= cnt;
                    if (room <= 0) break;
                    ++l2;
                    ++sub_bits;
                    room <<= 1;
                    cnt = count[l2];
                }
                if (used + (1u << sub_bits) > static_cast<uint32_t>(capacity)) return false;
                sub_base = used;
                used += 1u << sub_bits;
                for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[sub_base + i] = invalid;
                table[prefix] = make_entry(static_cast<uint32_t>(primary), sub_bits, kSubtable, sub_base);
            }
            const uint32_t e = symbol_entry(alphabet, sym, static_cast<uint32_t>(len - primary));
            for (uint32_t i = rev >> primary; i < (1u << sub_bits); i += 1u << (len - primary)) table[sub_base + i] = e;
        }
        // next code: +1 in the reversed domain, then append zeros when the length grows
        uint32_t incr = 1u << (len - 1);
        while (rev & incr) incr >>= 1;
        rev = incr ? (rev & (incr - 1)) + incr : 0;
        if (--remaining == 0 && idx + 1 < n_codes) {
            do ++len; while (!count[len]);
            remaining = count[len];
        }
    }
    return true;
}

constexpr int kLitCap = (1 << kLitBits) + 1024;  // subtables: <= 288 symbols, each <= 16 slots wide -> generous
constexpr int kDistCap = (1 << kDistBits) + 512;

struct FixedTables {
    uint32_t lit[kLitCap], dist[kDistCap];
    FixedTables() {
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(l, 288, 0, kLitBits, lit, kLitCap);
        build_table(d, 32, 1, kDistBits, dist, kDistCap);
    }
};

struct Bits {
    const uint8_t *in, *lim;  // lim: last position a refill may start from (the input copy is padded)
    uint64_t buf = 0;
    unsigned cnt = 0;
    bool refill() {  // afterwards cnt >= 56
        if (in > lim) return false;
        uint64_t w;
        memcpy(&w, in, 8);
        buf |= w << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
        return true;
    }
    uint32_t peek(unsigned n) const { return static_cast<uint32_t>(buf) & ((1u << n) - 1u); }
    void drop(unsigned n) {
        buf >>= n;
        cnt -= n;
    }
    uint32_t take(unsigned n) {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

inline uint32_t decode(const uint32_t *table, int primary, Bits &b) {  // one symbol, bits consumed; entry returned
    uint32_t e = table[b.peek(static_cast<unsigned>(primary))];
    if (e_kind(e) == kSubtable) {
        b.drop(e_bits(e));
        e = table[e_value(e) + b.peek(e_extra(e))];
    }
    b.drop(e_bits(e));
    return e;
}

struct Scratch {
    std::vector<uint8_t> in;  // padded copy of the compressed record
    uint32_t lit[kLitCap], dist[kDistCap], clen[1 << kClenBits];
};

// One DEFLATE stream being decoded, as a machine that advances in STEPS (a block header, or one trip of the symbol loop: up to
// three literals or one match).  A single stream runs its steps back to back.  TWO streams take turns (inflate_pair): the
// decoding of a symbol is one long dependence chain -- bit buffer -> table index -> load -> shift -> next index, 8-9 cycles on a
// core that could issue four instructions per cycle -- and two independent chains side by side fill what one leaves idle
// (measured on the fixture's records: 26.7 -> see profiles/r03_logs/host_inflate_pairs.log us per record and core).
struct Stream {
    enum State : int { kHeader = 0, kSymbols = 1, kDone = 2, kMalformed = -1, kFull = 3 };
    Bits b;
    const uint8_t *in_base = nullptr;
    size_t n = 0;
    uint8_t *out = nullptr, *o = nullptr, *oend = nullptr;
    const uint32_t *lit = nullptr, *dist = nullptr;
    Scratch *sc = nullptr;
    uint32_t final = 0;
    int state = kDone;

    void begin(const uint8_t *src, size_t len, uint8_t *dst, size_t cap, Scratch &scratch) {
        sc = &scratch;
        sc->in.resize(len + 16);
        memcpy(sc->in.data(), src, len);
        memset(sc->in.data() + len, 0, 16);
        in_base = sc->in.data();
        n = len;
        b = Bits();
        b.in = in_base;
        b.lim = in_base + len + 8;
        out = o = dst;
        oend = dst + cap;  // the caller leaves >= 8 bytes of slack behind oend
        final = 0;
        state = kHeader;
    }
    bool active() const { return state == kHeader || state == kSymbols; }

    // block header: stored blocks are copied whole; fixed / dynamic blocks set the tables up
    void header() {
        static const FixedTables fixed;
        if (!b.refill()) return fail();
        final = b.take(1);
        const uint32_t type = b.take(2);
        if (type == 0) {  // stored: back to the byte boundary, the bytes still sitting in the bit buffer are un-read
            b.drop(b.cnt & 7u);
            b.in -= b.cnt >> 3;
            b.buf = 0;
            b.cnt = 0;
            if (b.in + 4 > in_base + n) return fail();
            const uint32_t len = b.in[0] | (static_cast<uint32_t>(b.in[1]) << 8), nlen = b.in[2] | (static_cast<uint32_t>(b.in[3]) << 8);
            if ((len ^ nlen) != 0xffffu) return fail();
            b.in += 4;
            if (b.in + len > in_base + n) return fail();
            if (o + len > oend) {
                state = kFull;
                return;
            }
            memcpy(o, b.in, len);
            o += len;
            b.in += len;
            state = final ? kDone : kHeader;
            return;
        }
        if (type == 1) {
            lit = fixed.lit;
            dist = fixed.dist;
        } else if (type == 2) {
            const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return fail();
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (uint32_t i = 0; i < hclen; ++i) {
                if (b.cnt < 3 && !b.refill()) return fail();
                cl[order[i]] = static_cast<uint8_t>(b.take(3));
            }
            if (!build_table(cl, 19, 2, kClenBits, sc->clen, 1 << kClenBits)) return fail();
            uint8_t lens[286 + 30 + 138];
            uint32_t i = 0;
            while (i < hlit + hdist) {
                if (!b.refill()) return fail();
                const uint32_t e = sc->clen[b.peek(kClenBits)];
                if (e_kind(e) != kLiteral || e_bits(e) == 0) return fail();
                b.drop(e_bits(e));
                const uint32_t sym = e_value(e);
                if (sym < 16) {
                    lens[i++] = static_cast<uint8_t>(sym);
                } else if (sym == 16) {
                    if (i == 0) return fail();
                    const uint32_t rep = 3 + b.take(2);
                    memset(lens + i, lens[i - 1], rep);
                    i += rep;
                } else if (sym == 17) {
                    const uint32_t rep = 3 + b.take(3);
                    memset(lens + i, 0, rep);
                    i += rep;
                } else {
                    const uint32_t rep = 11 + b.take(7);
                    memset(lens + i, 0, rep);
                    i += rep;
                }
            }
            if (i != hlit + hdist || lens[256] == 0) return fail();
            if (!build_table(lens, static_cast<int>(hlit), 0, kLitBits, sc->lit, kLitCap)) return fail();
            if (!build_table(lens + hlit, static_cast<int>(hdist), 1, kDistBits, sc->dist, kDistCap)) return fail();
            lit = sc->lit;
            dist = sc->dist;
        } else {
            return fail();
        }
        state = kSymbols;
    }

    // one trip of the symbol loop: up to three literals, or a match, or the end of the block
    __attribute__((always_inline)) inline void symbols() {
        if (!b.refill()) return fail();
        if (o + 3 + 258 + 8 > oend) {  // room for three literals and one longest match plus the 8-byte copy overrun
            state = kFull;
            return;
        }
        uint32_t e = decode(lit, kLitBits, b);
        if (e_kind(e) == kLiteral) {  // up to three literals per refill (3 x 15 bits of the 56 available)
            *o++ = static_cast<uint8_t>(e_value(e));
            e = decode(lit, kLitBits, b);
            if (e_kind(e) == kLiteral) {
                *o++ = static_cast<uint8_t>(e_value(e));
                e = decode(lit, kLitBits, b);
                if (e_kind(e) == kLiteral) {
                    *o++ = static_cast<uint8_t>(e_value(e));
                    return;
                }
            }
            if (!b.refill()) return fail();  // up to 45 bits may be gone; a match needs up to 5 + 15 + 13 more
        }
        if (e_kind(e) == kEndOfBlock) {
            state = final ? kDone : kHeader;
            return;
        }
        if (e_kind(e) != kLength) return fail();
        const uint32_t len = e_value(e) + b.take(e_extra(e));
        const uint32_t de = decode(dist, kDistBits, b);
        if (e_kind(de) != kDistance) return fail();
        const uint32_t d = e_value(de) + b.take(e_extra(de));
        if (d > static_cast<size_t>(o - out)) return fail();
        const uint8_t *from = o - d;
        uint8_t *const stop = o + len;
        if (d >= 8) {
            do {
                memcpy(o, from, 8);  // chunks never overlap themselves: d >= 8
                o += 8;
                from += 8;
            } while (o < stop);
        } else {
            do {
                *o++ = *from++;
            } while (o < stop);
        }
        o = stop;
    }
    __attribute__((always_inline)) inline void step() {
        if (state == kSymbols)
            symbols();
        else
            header();
    }
    void fail() { state = kMalformed; }

    // 0 ok, 1 output full (grow and retry), -1 malformed; bytes produced / consumed
    int finish(size_t *produced, size_t *consumed) {
        if (state == kFull) return 1;
        if (state != kDone) return -1;
        b.in -= b.cnt >> 3;  // bytes not consumed but sitting in the bit buffer belong to what follows (the Adler-32 trailer)
        *produced = static_cast<size_t>(o - out);
        *consumed = static_cast<size_t>(b.in - in_base);
        return *consumed <= n ? 0 : -1;
    }
};

// returns 0 ok, 1 output full (grow and retry), -1 malformed
int inflate_raw(const uint8_t *src, size_t n, uint8_t *out, size_t cap, size_t *produced, size_t *consumed, Scratch &sc) {
    Stream s;
    s.begin(src, n, out, cap, sc);
    while (s.active()) s.step();
    return s.finish(produced, consumed);
}

// two streams, step by step in turn (see Stream); rc[0], rc[1] as inflate_raw's
void inflate_raw_pair(const uint8_t *const src[2], const size_t n[2], uint8_t *const out[2], const size_t cap[2], size_t produced[2],
                      size_t consumed[2], Scratch *const sc[2], int rc[2]) {
    Stream a, b;
    a.begin(src[0], n[0], out[0], cap[0], *sc[0]);
    b.begin(src[1], n[1], out[1], cap[1], *sc[1]);
    while (a.active() && b.active()) {
        a.step();
        b.step();
    }
    while (a.active()) a.step();
    while (b.active()) b.step();
    rc[0] = a.finish(&produced[0], &consumed[0]);
    rc[1] = b.finish(&produced[1], &consumed[1]);
}

}  // namespace

bool fast_inflate_zlib(const uint8_t *in, size_t n, std::vector<uint8_t> *out, size_t *len) {
    if (n < 6) return false;
    const uint32_t cmf = in[0], flg = in[1];
    if ((cmf & 15u) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20u)) return false;  // deflate, <= 32 KB window, no dictionary
    thread_local Scratch sc;
    if (out->size() < n * 4 + (1 << 16)) out->resize(n * 4 + (1 << 16));
    for (;;) {
        size_t produced = 0, consumed = 0;
        const int rc = inflate_raw(in + 2, n - 2, out->data(), out->size() - 8, &produced, &consumed, sc);
        if (rc < 0) return false;
        if (rc == 1) {
            if (out->size() > (size_t(1) << 31)) return false;
            out->resize(out->size() * 2);
            continue;
        }
        if (consumed + 4 > n - 2) return false;
        const uint8_t *t = in + 2 + consumed;
        const uint32_t want = (static_cast<uint32_t>(t[0]) << 24) | (static_cast<uint32_t>(t[1]) << 16) | (static_cast<uint32_t>(t[2]) << 8) | t[3];
        if (adler32_fast(out->data(), produced) != want) return false;
        *len = produced;
        return true;
    }
}

// Two streams at once (both must be well-formed zlib streams for the fast path; whichever is declined is reported per stream).
void fast_inflate_zlib_pair(const uint8_t *const in[2], const size_t n[2], std::vector<uint8_t> *const out[2], size_t len[2], bool ok[2]) {
    thread_local Scratch sc0, sc1;
    Scratch *const scs[2] = {&sc0, &sc1};
    bool head_ok[2];
    for (int k = 0; k < 2; ++k) {
        ok[k] = false;
        head_ok[k] = false;
        if (n[k] < 6) continue;
        const uint32_t cmf = in[k][0], flg = in[k][1];
        if ((cmf & 15u) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20u)) continue;
        head_ok[k] = true;
        if (out[k]->size() < n[k] * 4 + (1 << 16)) out[k]->resize(n[k] * 4 + (1 << 16));
    }
    if (!(head_ok[0] && head_ok[1])) {  // nothing to pair: each through the single-stream decoder
        for (int k = 0; k < 2; ++k) ok[k] = head_ok[k] && fast_inflate_zlib(in[k], n[k], out[k], &len[k]);
        return;
    }
    const uint8_t *const src[2] = {in[0] + 2, in[1] + 2};
    const size_t sn[2] = {n[0] - 2, n[1] - 2};
    uint8_t *const dst[2] = {out[0]->data(), out[1]->data()};
    const size_t cap[2] = {out[0]->size() - 8, out[1]->size() - 8};
    size_t produced[2] = {0, 0}, consumed[2] = {0, 0};
    int rc[2];
    inflate_raw_pair(src, sn, dst, cap, produced, consumed, scs, rc);
    for (int k = 0; k < 2; ++k) {
        if (rc[k] == 1) {  // output did not fit (rare: > 4x + 64 KB): the single-stream decoder grows and retries
            ok[k] = fast_inflate_zlib(in[k], n[k], out[k], &len[k]);
            continue;
        }
        if (rc[k] < 0 || consumed[k] + 4 > sn[k]) continue;
        const uint8_t *t = src[k] + consumed[k];
        const uint32_t want = (static_cast<uint32_t>(t[0]) << 24) | (static_cast<uint32_t>(t[1]) << 16) | (static_cast<uint32_t>(t[2]) << 8) | t[3];
        if (adler32_fast(dst[k], produced[k]) != want) continue;
        len[k] = produced[k];
        ok[k] = true;
    }
}

}  // namespace sfa
