// blow5_kernels.hpp -- BLOW5 records decoded ON THE DEVICE (SURVEY.md section 8f-2 formats, 8f-1 overlap): the record
// decompressor (zlib / DEFLATE, RFC 1950 / 1951) and the signal decompressor (StreamVByte of zig-zag deltas, "svb-zd") of
// the files real runs produce, as gfx950 kernels.  What the reference does per read on a host thread
// (slow5_rec_depress_parse, slow5lib/src/slow5.c:2575-2593 inflate of the whole record, 2806-2925 primary fields,
// slow5lib/src/slow5_press.c:1085-1135 + thirdparty/streamvbyte svb-zd) is 27 us of inflate and 3.5 us of StreamVByte per
// read on one core here -- 88 % of the host stage of a compressed run, and the reason the command line reached 0.42 M
// reads/s on compressed files against 0.68 M on uncompressed ones.  On the device:
//
//   * blow5_inflate_kernel: ONE LANE PER RECORD.  DEFLATE is a serial bit stream; there is nothing to share between the
//     lanes of a wave, but a batch has thousands of records and a record is only a few KB.  Each lane keeps its canonical
//     Huffman description (count per code length + symbols in code order, as RFC 1951 3.2.2 defines the code) and a
//     9-bit first-level table of the literal/length code in LDS (1.8 KB per lane: one 64-lane block per CU), its bit
//     buffer in registers.  Stored, fixed and dynamic blocks; the Adler-32 trailer is checked.  Anything malformed marks
//     the record failed (length -1) and the caller falls back to the host reader for the batch.
//   * blow5_fields_kernel: the primary fields of every inflated record (read id, digitisation, offset, range, sample
//     count; slow5.c:2806-2925) into fixed-size rows for the host, which needs them for the output lines.
//   * blow5_svb_kernel: ONE WAVE PER RECORD.  Lane l of round r owns key byte 64 r + l, i.e. four values whose byte
//     lengths are the key's four 2-bit fields; a wave scan of the lanes' byte totals gives every lane its place in the data
//     stream, a second scan of the zig-zag decoded deltas gives the samples, stored 8 bytes per lane, coalesced.
#pragma once

#ifdef SFA_HOST_HARNESS  // tests/c/device_inflate_host.cpp compiles the lane decoder below with g++ (sanitizers, against zlib)
#include <string.h>
#define __device__
#define __forceinline__ inline
static inline bool __any(bool x) { return x; }
static inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i, x >>= 1) r = (r << 1) | (x & 1u);
    return r;
}
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace sfa {

struct InflateArgs {
    const uint8_t *in;       // records back to back (8 readable bytes of padding behind the last one)
    const int64_t *in_off;   // [n+1]
    uint8_t *out;            // inflated payloads, slot i = [out_off[i], out_off[i+1])
    const int64_t *out_off;  // [n+1] capacities
    int32_t *out_len;        // [n] bytes produced, -1: failed (malformed, truncated, or larger than its slot)
    int32_t n;
};

constexpr int kInfFastBits = 9;
// per-lane decoder state in LDS
struct InflateLds {
    uint16_t fast[1 << kInfFastBits];  // literal/length code, first level: (symbol << 4) | code length, 0 = longer than 9 bits
    uint16_t lcount[16];               // literal/length code: codes per length ...
    uint16_t lsym[288];                // ... and symbols in code order
    uint16_t dcount[16];               // distance code
    uint16_t dsym[32];
    uint16_t ccount[8];                // code-length code (dynamic block header)
    uint16_t csym[20];
};

__device__ const uint16_t kInfLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kInfLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kInfDistBase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                             193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kInfDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kInfClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// LSB-first bit reader over a byte range.  Input arrives 16 bytes at a time into a 128-bit shift register (lo, hi) with the NEXT
// 16 bytes already requested (nlo, nhi): a load is issued sixteen-odd symbols before its first bit is looked at.  The buffer
// is padded so that reading ahead is harmless; `avail` keeps the books on what was really part of the record.
// (Measured and dropped, profiles/r02_logs/blow5_decode_variants.log: a 32-byte window per lane reloaded by all lanes of the
// wave together, and 8-byte output stores -- both meant to spare the wave memory waits, both slower than this: what a wave
// of independent decoders pays for is INSTRUCTIONS, every lane's rare path being executed by all of them; see
// blow5_inflate_kernel for what does help.)
struct BitReader {
    const uint8_t *p;   // next 16 bytes to request
    int64_t avail;      // bits of the record not yet moved into buf
    uint64_t buf;
    int cnt;            // valid bits in buf (may include bits beyond the record once avail < 0: see drop())
    uint64_t lo, hi;    // words waiting to enter buf
    uint64_t nlo, nhi;  // the 16 bytes behind them, in flight
    int words;          // 32-bit words left in (lo, hi)
    bool overrun;
    __device__ __forceinline__ static void load16(const uint8_t *q, uint64_t &a, uint64_t &b) {
        uint64_t t[2];
        __builtin_memcpy(t, q, 16);
        a = t[0];
        b = t[1];
    }
    __device__ __forceinline__ void init(const uint8_t *b, int64_t nbytes) {
        avail = nbytes * 8;
        buf = 0;
        cnt = 0;
        overrun = false;
        load16(b, lo, hi);
        load16(b + 16, nlo, nhi);
        p = b + 32;
        words = 4;
    }
    __device__ __forceinline__ void sync() {}  // (hook of the dropped window variant; kept so that the loops read the same)
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            buf |= (lo & 0xffffffffull) << cnt;
            lo = (lo >> 32) | (hi << 32);
            hi >>= 32;
            cnt += 32;
            avail -= 32;
            if (--words == 0) {
                lo = nlo;
                hi = nhi;
                load16(p, nlo, nhi);
                p += 16;
                words = 4;
            }
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return static_cast<uint32_t>(buf) & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) {
        buf >>= n;
        cnt -= n;
        if (avail < 0 && cnt < -avail) overrun = true;  // consumed bits that were never part of the record
    }
    __device__ __forceinline__ uint32_t get(int n) {  // n <= 16
        refill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    __device__ __forceinline__ void align_byte() { drop(cnt & 7); }
};

// canonical decode, one bit at a time (RFC 1951 3.2.2; the loop of Mark Adler's puff.c written down from the RFC's rules):
// codes of each length are consecutive integers, the first code of a length follows from the counts of the shorter ones
__device__ __forceinline__ int inf_decode_slow(BitReader &br, const uint16_t *count, const uint16_t *sym, int maxlen) {
    br.refill();
    int code = 0, first = 0, index = 0;
    uint32_t bits = static_cast<uint32_t>(br.buf);
    for (int len = 1; len <= maxlen; ++len) {
        code |= bits & 1;
        bits >>= 1;
        const int c = count[len];
        if (code - c < first) {
            br.drop(len);
            return sym[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// counts + symbols in code order from code lengths; false on an over-subscribed code and on the incomplete codes zlib's
// inflate refuses (so that the device accepts exactly what the reference's reader accepts): an incomplete code is only
// legal with a single code of length 1 (one distance used); `no_codes_ok`: an empty code (a block without matches);
// `any`: the fixed distance code of RFC 1951 3.2.6, which has 30 of 32 codes by definition
__device__ __forceinline__ bool inf_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *sym, int maxlen, bool no_codes_ok, bool any = false) {
    for (int l = 0; l <= maxlen; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    if (count[0] == n) return no_codes_ok;  // no codes at all
    int left = 1;
    for (int l = 1; l <= maxlen; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    if (left > 0 && !any) {
        int used = 0;
        for (int l = 1; l <= maxlen; ++l) used += count[l];
        if (!(used == 1 && count[1] == 1)) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < maxlen; ++l) offs[l + 1] = offs[l] + count[l];
    for (int s = 0; s < n; ++s)
        if (lens[s]) sym[offs[lens[s]]++] = static_cast<uint16_t>(s);
    count[0] = 0;
    return true;
}

__device__ __forceinline__ void inf_build_fast(const uint16_t *count, const uint16_t *sym, uint16_t *fast) {
    for (int i = 0; i < (1 << kInfFastBits); ++i) fast[i] = 0;
    int code = 0, index = 0;
    for (int len = 1; len <= kInfFastBits; ++len) {
        for (int k = 0; k < count[len]; ++k) {
            // DEFLATE sends codes most significant bit first into an LSB-first stream: the table index is the reversed code
            uint32_t r = __brev(static_cast<uint32_t>(code)) >> (32 - len);
            const uint16_t e = static_cast<uint16_t>((sym[index] << 4) | len);
            for (uint32_t i = r; i < (1u << kInfFastBits); i += (1u << len)) fast[i] = e;
            ++code;
            ++index;
        }
        code <<= 1;
    }
}

// one zlib stream -> bytes; returns bytes produced or -1
__device__ inline int inflate_zlib_lane(const uint8_t *in, int64_t n_in, uint8_t *out, int64_t cap, InflateLds &S) {
    if (n_in < 6) return -1;
    const uint32_t cmf = in[0], flg = in[1];
    if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return -1;  // deflate, header check, no preset dictionary
    BitReader br;
    br.init(in + 2, n_in - 2 - 4);
    int64_t op = 0;
    uint32_t s1 = 1, s2 = 0;  // Adler-32 over the output, folded in as bytes are produced (sums stay below 2^32 for 5552 bytes)
    int since_mod = 0;
    auto emit = [&](uint8_t b) {
        out[op++] = b;
        s1 += b;
        s2 += s1;
        if (++since_mod == 5552) {
            s1 %= 65521u;
            s2 %= 65521u;
            since_mod = 0;
        }
    };
    uint8_t *lens = reinterpret_cast<uint8_t *>(S.fast);  // the first-level table's storage doubles as scratch for the code lengths
    for (;;) {
        br.sync();
        const uint32_t last = br.get(1), type = br.get(2);
        if (type == 0) {  // stored
            br.align_byte();
            const uint32_t len = br.get(16), nlen = br.get(16);
            if ((len ^ 0xffffu) != nlen || op + len > cap) return -1;
            for (uint32_t i = 0; i < len; ++i) {
                br.sync();
                emit(static_cast<uint8_t>(br.get(8)));
                if (br.overrun) return -1;  // (every loop checks: a corrupt length must not walk far beyond the record)
            }
        } else if (type == 1 || type == 2) {
            int nlit, ndist;
            if (type == 1) {  // fixed code, RFC 1951 3.2.6
                for (int s = 0; s < 144; ++s) lens[s] = 8;
                for (int s = 144; s < 256; ++s) lens[s] = 9;
                for (int s = 256; s < 280; ++s) lens[s] = 7;
                for (int s = 280; s < 288; ++s) lens[s] = 8;
                for (int s = 0; s < 30; ++s) lens[288 + s] = 5;
                nlit = 288;
                ndist = 30;
            } else {  // dynamic code, 3.2.7
                nlit = br.get(5) + 257;
                ndist = br.get(5) + 1;
                const int ncl = br.get(4) + 4;
                if (nlit > 286 || ndist > 30) return -1;
                uint8_t cl[19];
                for (int i = 0; i < 19; ++i) cl[i] = 0;
                br.sync();
                for (int i = 0; i < ncl; ++i) {
                    if ((i & 7) == 7) br.sync();
                    cl[kInfClOrder[i]] = static_cast<uint8_t>(br.get(3));
                }
                if (!inf_build(cl, 19, S.ccount, S.csym, 7, false)) return -1;
                int i = 0;
                while (i < nlit + ndist) {
                    br.sync();
                    const int sym = inf_decode_slow(br, S.ccount, S.csym, 7);
                    if (sym < 0 || br.overrun) return -1;
                    if (sym < 16) {
                        lens[i++] = static_cast<uint8_t>(sym);
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) return -1;
                            val = lens[i - 1];
                            rep = 3 + br.get(2);
                        } else if (sym == 17) {
                            rep = 3 + br.get(3);
                        } else {
                            rep = 11 + br.get(7);
                        }
                        if (i + rep > nlit + ndist) return -1;
                        while (rep--) lens[i++] = static_cast<uint8_t>(val);
                    }
                }
                if (lens[256] == 0) return -1;  // no end-of-block code
                // (the distance lengths follow the literal/length ones in `lens`; move them to where the fixed case has them)
                for (int s = ndist - 1; s >= 0; --s) lens[288 + s] = lens[nlit + s];
            }
            if (!inf_build(lens, nlit, S.lcount, S.lsym, 15, false)) return -1;
            if (!inf_build(lens + 288, ndist, S.dcount, S.dsym, 15, true, type == 1)) return -1;
            inf_build_fast(S.lcount, S.lsym, S.fast);  // (overwrites `lens`: no longer needed)
            for (;;) {
                br.sync();
                br.refill();
                int sym;
                const uint32_t e = S.fast[br.peek(kInfFastBits)];
                if (e & 15u) {
                    br.drop(e & 15u);
                    sym = e >> 4;
                } else {
                    sym = inf_decode_slow(br, S.lcount, S.lsym, 15);
                    if (sym < 0) return -1;
                }
                if (sym < 256) {
                    if (op >= cap) return -1;
                    emit(static_cast<uint8_t>(sym));
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29) return -1;
                    const int len = kInfLenBase[sym] + static_cast<int>(br.get(kInfLenExtra[sym]));
                    const int ds = inf_decode_slow(br, S.dcount, S.dsym, 15);
                    if (ds < 0 || ds >= 30) return -1;
                    const int xb = kInfDistExtra[ds];
                    int dist = kInfDistBase[ds];
                    if (xb) {
                        br.refill();
                        dist += static_cast<int>(br.peek(xb));
                        br.drop(xb);
                    }
                    if (dist > op || op + len > cap) return -1;
                    for (int k = 0; k < len; ++k) emit(out[op - dist]);
                }
                if (br.overrun) return -1;
            }
        } else {
            return -1;
        }
        if (br.overrun) return -1;
        if (last) break;
    }
    // trailer: Adler-32 of the output, big endian, in the four bytes behind the deflate data
    s1 %= 65521u;
    s2 %= 65521u;
    const uint8_t *t = in + n_in - 4;
    const uint32_t want = (static_cast<uint32_t>(t[0]) << 24) | (static_cast<uint32_t>(t[1]) << 16) | (static_cast<uint32_t>(t[2]) << 8) | t[3];
    if (want != ((s2 << 16) | s1)) return -1;
    return static_cast<int>(op);
}

#ifdef SFA_DEFINE_BLOW5_KERNELS  // plain kernels: defined in exactly one translation unit (sfa_pre.hip)
// `lanes` records per wave (one wave per block, the other lanes idle), chosen by the host from the batch size
// (inflate_lanes()).  A wave of independent decoders executes every lane's path: whenever ONE lane meets a code longer than
// the first-level table, a match or a block header, all of them sit through it -- with 32 records per wave that is nearly
// every iteration, and a batch took 6.5-10 ms whatever its size (one wave's serial time).  A batch of a few thousand records
// is far too small to fill 1 024 SIMDs with full waves anyway, so the records are spread thin instead: enough waves for
// about one per SIMD, down to ONE record per wave (no divergence at all), up to kInfMaxLanes for the largest batches.
constexpr int kInfMaxLanes = 32;
__global__ void __launch_bounds__(64) blow5_inflate_kernel(const InflateArgs a, const int lanes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (static_cast<int>(threadIdx.x) >= lanes) return;
    InflateLds *S = reinterpret_cast<InflateLds *>(lds_raw) + threadIdx.x;
    const int i = blockIdx.x * lanes + threadIdx.x;
    if (i >= a.n) return;
    const int64_t b = a.in_off[i], e = a.in_off[i + 1];
    const int64_t ob = a.out_off[i], oe = a.out_off[i + 1];
    a.out_len[i] = inflate_zlib_lane(a.in + b, e - b, a.out + ob, oe - ob, *S);
}

// Primary fields of a record (slow5.c:2806-2925): u16 id_len, id, u32 read_group, f64 digitisation, offset, range,
// sampling_rate, u64 len, signal.  One lane per record; `head` gets a fixed-size row for the host:
//   [0] i32 status (0 ok, -1 malformed)  [4] i32 id_len  [8] i64 n_samples  [16] f64 x3 digitisation, offset, range
//   [40] i64 signal offset inside the payload  [48] i64 signal bytes  [56..] the read id (up to kBlow5IdMax bytes)
constexpr int kBlow5HeadBytes = 192;
constexpr int kBlow5IdMax = kBlow5HeadBytes - 56;
struct FieldsArgs {
    const uint8_t *payload;      // inflated payloads (or the records themselves when they are not compressed)
    const int64_t *payload_off;  // [n+1] slots
    const int32_t *payload_len;  // [n] valid bytes in each slot (nullptr: the whole slot)
    uint8_t *head;               // [n][kBlow5HeadBytes]
    int32_t signal_svb;          // 1: the signal is svb-zd (u32 n + keys + data), 0: plain int16
    int32_t n;
};

__global__ void __launch_bounds__(64) blow5_fields_kernel(const FieldsArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    uint8_t *h = a.head + static_cast<int64_t>(i) * kBlow5HeadBytes;
    const uint8_t *p = a.payload + a.payload_off[i];
    const int64_t len = a.payload_len ? a.payload_len[i] : (a.payload_off[i + 1] - a.payload_off[i]);
    int32_t status = 0, id_len = 0;
    int64_t n_samples = 0, sig_off = 0, sig_bytes = 0;
    double f[3] = {0, 0, 0};
    if (len < 2) {
        status = -1;
    } else {
        id_len = p[0] | (p[1] << 8);
        const int64_t fixed = 2 + id_len + 4 + 8 * 4 + 8;
        if (id_len > kBlow5IdMax || len < fixed) {
            status = -1;
        } else {
            for (int k = 0; k < id_len; ++k) h[56 + k] = p[2 + k];
            const uint8_t *q = p + 2 + id_len + 4;
            __builtin_memcpy(&f[0], q, 8);
            __builtin_memcpy(&f[1], q + 8, 8);
            __builtin_memcpy(&f[2], q + 16, 8);
            uint64_t l;
            __builtin_memcpy(&l, q + 32, 8);
            sig_off = fixed;
            if (a.signal_svb) {  // l = compressed bytes; the sample count is the u32 in front of the keys
                sig_bytes = static_cast<int64_t>(l);
                if (sig_bytes < 4 || fixed + sig_bytes > len) {
                    status = -1;
                } else {
                    uint32_t ns;
                    __builtin_memcpy(&ns, p + fixed, 4);
                    n_samples = ns;
                    const int64_t keys = (n_samples + 3) / 4;
                    if (4 + keys > sig_bytes || n_samples > (int64_t(1) << 31)) status = -1;
                }
            } else {
                n_samples = static_cast<int64_t>(l);
                sig_bytes = 2 * n_samples;
                if (n_samples < 0 || fixed + sig_bytes > len) status = -1;
            }
        }
    }
    __builtin_memcpy(h, &status, 4);
    __builtin_memcpy(h + 4, &id_len, 4);
    __builtin_memcpy(h + 8, &n_samples, 8);
    __builtin_memcpy(h + 16, f, 24);
    __builtin_memcpy(h + 40, &sig_off, 8);
    __builtin_memcpy(h + 48, &sig_bytes, 8);
}

// svb-zd -> int16 samples (slow5_press.c:1085-1135: StreamVByte with 1-4 byte values, zig-zag, delta against the previous
// sample, first sample against 0; the reference stores the 32-bit results as int16).  One wave per record.
struct SvbArgs {
    const uint8_t *payload;
    const int64_t *payload_off;  // [n+1]
    const uint8_t *head;         // rows of blow5_fields_kernel
    const int64_t *raw_off;      // [n+1] sample offsets of the output
    int16_t *raw;
    int32_t *bad;                // [n] set to 1 when the data stream is shorter than its keys say
    int32_t signal_svb;
    int32_t n;
};

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total) {
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(s, d);
        if (lane >= d) s += t;
    }
    *total = __shfl(s, 63);
    return s - v;
}

__global__ void __launch_bounds__(256) blow5_svb_kernel(const SvbArgs a) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n) return;
    const int lane = threadIdx.x & 63;
    const uint8_t *h = a.head + static_cast<int64_t>(i) * kBlow5HeadBytes;
    int32_t status;
    int64_t n_samples, sig_off, sig_bytes;
    __builtin_memcpy(&status, h, 4);
    __builtin_memcpy(&n_samples, h + 8, 8);
    __builtin_memcpy(&sig_off, h + 40, 8);
    __builtin_memcpy(&sig_bytes, h + 48, 8);
    if (status != 0) return;
    int16_t *out = a.raw + a.raw_off[i];
    const uint8_t *sig = a.payload + a.payload_off[i] + sig_off;
    if (!a.signal_svb) {  // plain int16 samples (unaligned in the payload)
        for (int64_t j = lane; j < n_samples; j += 64) {
            int16_t v;
            __builtin_memcpy(&v, sig + 2 * j, 2);
            out[j] = v;
        }
        return;
    }
    const int64_t n_keys = (n_samples + 3) / 4;
    const uint8_t *keys = sig + 4, *data = keys + n_keys;
    const int64_t data_bytes = sig_bytes - 4 - n_keys;
    int64_t dpos = 0;  // bytes of the data stream consumed by the rounds so far
    int prev = 0;      // last sample of the previous round
    bool short_data = false;
    for (int64_t k0 = 0; k0 < n_keys; k0 += 64) {
        const int64_t k = k0 + lane;
        const uint32_t key = k < n_keys ? keys[k] : 0;
        // (the last key byte may describe fewer than four values: the missing ones have no bytes in the data stream)
        const int64_t left = n_samples - 4 * k;
        const int l0 = left > 0 ? (key & 3) + 1 : 0, l1 = left > 1 ? ((key >> 2) & 3) + 1 : 0, l2 = left > 2 ? ((key >> 4) & 3) + 1 : 0,
                  l3 = left > 3 ? ((key >> 6) & 3) + 1 : 0;
        const int mine = l0 + l1 + l2 + l3;
        int total;
        const int64_t at = dpos + wave_excl_scan(mine, lane, &total);
        uint32_t v[4] = {0, 0, 0, 0};
        if (k < n_keys) {
            if (at + mine > data_bytes) {
                short_data = true;
            } else {
                uint8_t b[16];
                for (int t = 0; t < 16; ++t) b[t] = t < mine ? data[at + t] : 0;  // (unrolled byte loads: L1 hits, the 64 lanes read ~600 consecutive bytes)
                int o = 0;
                const int ls[4] = {l0, l1, l2, l3};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t x = 0;
                    for (int t = 0; t < ls[q]; ++t) x |= static_cast<uint32_t>(b[o + t]) << (8 * t);
                    v[q] = x;
                    o += ls[q];
                }
            }
        }
        // zig-zag -> delta, running sum inside the lane, then across the lanes
        int d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = static_cast<int>((v[q] >> 1) ^ (0u - (v[q] & 1u)));
        const int64_t base = 4 * k;
        // values past n_samples (the last key byte may describe fewer than four) must not enter the sums
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (base + q >= n_samples) d[q] = 0;
        const int s0 = d[0], s1 = s0 + d[1], s2 = s1 + d[2], s3 = s2 + d[3];
        int tot;
        const int before = prev + wave_excl_scan(s3, lane, &tot);
        const int o4[4] = {before + s0, before + s1, before + s2, before + s3};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (base + q < n_samples) out[base + q] = static_cast<int16_t>(o4[q]);
        prev += tot;
        dpos += total;
    }
    if (__any(short_data) && lane == 0) a.bad[i] = 1;
}
#endif

}  // namespace sfa
