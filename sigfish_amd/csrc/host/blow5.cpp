// blow5.cpp -- see blow5.hpp.  zlib is the only dependency (zstd-compressed files are rejected with a message,
// like a reference binary built without zstd=1).
#include "blow5.hpp"

#include "inflate.hpp"

#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace sfa {

namespace {
const unsigned char kMagic[6] = {'B', 'L', 'O', 'W', '5', 1};
const unsigned char kEof[5] = {'5', 'W', 'O', 'L', 'B'};
const char kAsciiFirst[] = "#slow5_version\t";
const char kAsciiGroups[] = "#num_read_groups\t";
const char kAsciiTypes[] = "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*";
const char kAsciiNames[] = "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal";

template <typename T>
bool take(const uint8_t *&p, const uint8_t *end, T *out) {
    if (static_cast<size_t>(end - p) < sizeof(T)) return false;
    memcpy(out, p, sizeof(T));
    p += sizeof(T);
    return true;
}

// One zlib stream and one output buffer per thread, reused for every record (inflateInit2 allocates ~40 KB and a fresh
// vector zero-fills its 64 KB: together a fifth of the cost of decoding a 4 KB record).  The buffer only grows and
// keeps its size; *len receives the number of bytes produced.
struct Inflater {
    z_stream zs;
    bool ready = false;
    std::vector<uint8_t> buf;
    ~Inflater() {
        if (ready) inflateEnd(&zs);
    }
};

bool inflate_all(const uint8_t *src, size_t n, const uint8_t **out, size_t *len) {
    thread_local Inflater inf;
    if (fast_inflate_zlib(src, n, &inf.buf, len)) {  // own decoder (inflate.hpp); anything it declines goes to zlib
        *out = inf.buf.data();
        return true;
    }
    if (!inf.ready) {
        memset(&inf.zs, 0, sizeof inf.zs);
        if (inflateInit2(&inf.zs, MAX_WBITS) != Z_OK) return false;
        inf.ready = true;
        inf.buf.resize(1 << 16);
    } else if (inflateReset(&inf.zs) != Z_OK) {
        return false;
    }
    if (inf.buf.size() < n * 4) inf.buf.resize(n * 4);
    z_stream &zs = inf.zs;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = static_cast<uInt>(n);
    size_t have = 0;
    int rc;
    do {
        if (have == inf.buf.size()) inf.buf.resize(inf.buf.size() * 2);
        zs.next_out = inf.buf.data() + have;
        zs.avail_out = static_cast<uInt>(inf.buf.size() - have);
        rc = inflate(&zs, Z_NO_FLUSH);
        have = inf.buf.size() - zs.avail_out;
    } while (rc == Z_OK);
    if (rc != Z_STREAM_END) return false;
    *out = inf.buf.data();
    *len = have;
    return true;
}

// StreamVByte + zig-zag delta -> int16 samples.  Key byte k holds the byte counts (minus one) of four values; the
// SSSE3 path expands them with one pshufb per key byte (shuffle table built once), un-zigzags and prefix-sums four
// deltas at a time; the scalar loop handles the tail and machines without SSSE3.
struct SvbTables {
    uint8_t shuf[256][16];
    uint8_t len[256];
    SvbTables() {
        for (int k = 0; k < 256; ++k) {
            int pos = 0;
            for (int v = 0; v < 4; ++v) {
                const int nb = ((k >> (2 * v)) & 3) + 1;
                for (int b = 0; b < 4; ++b) shuf[k][4 * v + b] = b < nb ? static_cast<uint8_t>(pos + b) : 0x80;  // 0x80 -> zero byte
                pos += nb;
            }
            len[k] = static_cast<uint8_t>(pos);
        }
    }
};
const SvbTables &svb_tables() {
    static const SvbTables t;
    return t;
}

__attribute__((target("ssse3"))) size_t decode_svb_zd_ssse3(const uint8_t *keys, const uint8_t **datap, const uint8_t *end, uint32_t n,
                                                            int16_t *out, int32_t *prevp) {
    const SvbTables &t = svb_tables();
    const uint8_t *data = *datap;
    __m128i prev = _mm_set1_epi32(*prevp);
    size_t i = 0;
    // 16 readable bytes are needed per group; the last groups go through the scalar loop
    for (; i + 4 <= n && static_cast<size_t>(end - data) >= 16; i += 4) {
        const unsigned k = keys[i >> 2];
        const __m128i raw = _mm_loadu_si128(reinterpret_cast<const __m128i *>(data));
        const __m128i v = _mm_shuffle_epi8(raw, _mm_loadu_si128(reinterpret_cast<const __m128i *>(t.shuf[k])));
        data += t.len[k];
        // zig-zag: (v >> 1) ^ -(v & 1)
        const __m128i d = _mm_xor_si128(_mm_srli_epi32(v, 1), _mm_sub_epi32(_mm_setzero_si128(), _mm_and_si128(v, _mm_set1_epi32(1))));
        // inclusive prefix sum of the four deltas on top of the running value
        __m128i s = _mm_add_epi32(d, _mm_slli_si128(d, 4));
        s = _mm_add_epi32(s, _mm_slli_si128(s, 8));
        s = _mm_add_epi32(s, prev);
        prev = _mm_shuffle_epi32(s, 0xff);
        alignas(16) int32_t tmp[4];
        _mm_store_si128(reinterpret_cast<__m128i *>(tmp), s);
        out[i] = static_cast<int16_t>(tmp[0]);
        out[i + 1] = static_cast<int16_t>(tmp[1]);
        out[i + 2] = static_cast<int16_t>(tmp[2]);
        out[i + 3] = static_cast<int16_t>(tmp[3]);
    }
    *prevp = _mm_cvtsi128_si32(prev);
    *datap = data;
    return i;
}

bool decode_svb_zd(const uint8_t *p, size_t nbytes, std::vector<int16_t> *out) {
    if (nbytes < 4) return false;
    uint32_t n;
    memcpy(&n, p, 4);
    const uint8_t *keys = p + 4;
    const size_t nkeys = (static_cast<size_t>(n) + 3) / 4;
    if (4 + nkeys > nbytes) return false;
    const uint8_t *data = keys + nkeys;
    const uint8_t *end = p + nbytes;
    out->resize(n);
    int32_t prev = 0;
    size_t i = 0;
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3");
    if (have_ssse3) i = decode_svb_zd_ssse3(keys, &data, end, n, out->data(), &prev);
    for (; i < n; ++i) {
        const unsigned code = (keys[i >> 2] >> ((i & 3) * 2)) & 3u;  // bytes-1 of value i
        if (static_cast<size_t>(end - data) < code + 1) return false;
        uint32_t v = 0;
        memcpy(&v, data, code + 1);  // little endian
        data += code + 1;
        const int32_t delta = static_cast<int32_t>(v >> 1) ^ -static_cast<int32_t>(v & 1);
        (*out)[i] = static_cast<int16_t>(delta + prev);
        prev += delta;
    }
    return data == end;
}
}  // namespace

void Blow5Reader::stop_prefault() {
    if (prefault_.joinable()) {
        prefault_quit_ = true;
        prefault_.join();
    }
    prefault_quit_ = false;
}

void Blow5Reader::start_prefault(size_t window) {
    stop_prefault();
    if (!map_) return;
    consumed_ = map_pos_;
    const uint8_t *base = map_;
    // (a shard's helper stops a page past the shard: the next rank's pages are the next rank's business)
    const size_t size = limit_pos_ < map_size_ ? std::min<size_t>(map_size_, static_cast<size_t>(limit_pos_) + 4096) : map_size_, start = map_pos_;
    prefault_ = std::thread([this, base, size, start, window] {
        unsigned sink = 0;
        for (size_t pos = start & ~size_t(4095); pos < size && !prefault_quit_; pos += 4096) {
            while (pos > consumed_.load(std::memory_order_relaxed) + window && !prefault_quit_) {
                struct timespec ts = {0, 200000};  // 0.2 ms: far ahead already
                nanosleep(&ts, nullptr);
            }
            sink += *reinterpret_cast<const volatile uint8_t *>(base + pos);
        }
        (void)sink;
    });
}

void Blow5Reader::close() {
    stop_prefault();
    if (map_) munmap(const_cast<uint8_t *>(map_), map_size_);
    map_ = nullptr;
    map_size_ = map_pos_ = 0;
    if (fp_) fclose(fp_);
    fp_ = nullptr;
}

const char *Blow5Reader::attr(const std::string &key) const {
    auto it = attrs_.find(key);
    return it == attrs_.end() ? nullptr : it->second.c_str();
}

bool Blow5Reader::open(const std::string &path) {
    close();
    attrs_.clear();
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) {
        err_ = "cannot open " + path;
        return false;
    }
    ascii_ = false;
    n_aux_ = 0;
    eof_bytes_ = sizeof kEof;
    unsigned char head[68];
    const size_t got_head = fread(head, 1, sizeof head, fp_);
    if (got_head >= sizeof kAsciiFirst - 1 && memcmp(head, kAsciiFirst, sizeof kAsciiFirst - 1) == 0) return open_ascii(path);
    if (got_head != sizeof head || memcmp(head, kMagic, sizeof kMagic) != 0) {
        err_ = path + ": not a BLOW5 file (bad magic number)";
        return false;
    }
    const unsigned major = head[6], minor = head[7];
    record_press_ = head[9];
    memcpy(&n_groups_, head + 10, 4);
    signal_press_ = (major > 0 || minor >= 2) ? head[14] : 0;
    uint32_t hsize;
    memcpy(&hsize, head + 64, 4);
    {  // sizes read from the file are only trusted up to the size of the file
        struct stat fsb;
        file_size_ = (fstat(fileno(fp_), &fsb) == 0 && S_ISREG(fsb.st_mode)) ? static_cast<uint64_t>(fsb.st_size) : UINT64_MAX;
    }
    if (hsize > file_size_) {
        err_ = path + ": malformed BLOW5 header (its size field exceeds the file)";
        return false;
    }
    std::string text(hsize, '\0');
    if (hsize && fread(&text[0], 1, hsize, fp_) != hsize) {
        err_ = path + ": truncated BLOW5 header";
        return false;
    }
    data_begin_ = pos_ = 68 + static_cast<uint64_t>(hsize);
    limit_pos_ = limit_records_ = UINT64_MAX;
    if (record_press_ > 1) {
        err_ = path + ": record compression method " + std::to_string(record_press_) + " is not supported (zlib or none only)";
        return false;
    }
    if (signal_press_ > 1) {
        err_ = path + ": signal compression method " + std::to_string(signal_press_) + " is not supported (svb-zd or none only)";
        return false;
    }
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        if (text[pos] == '@') {
            const size_t tab = text.find('\t', pos);
            if (tab != std::string::npos && tab < eol) {
                size_t vend = text.find('\t', tab + 1);  // value of read group 0
                if (vend == std::string::npos || vend > eol) vend = eol;
                attrs_[text.substr(pos + 1, tab - pos - 1)] = text.substr(tab + 1, vend - tab - 1);
            }
        }
        pos = eol + 1;
    }
    // map the file for the zero-copy record iterator; plain fread keeps working if this fails (pipes, odd filesystems)
    struct stat sb;
    if (fstat(fileno(fp_), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
        void *m = mmap(nullptr, static_cast<size_t>(sb.st_size), PROT_READ, MAP_PRIVATE, fileno(fp_), 0);
        if (m != MAP_FAILED) {
            map_ = static_cast<const uint8_t *>(m);
            map_size_ = static_cast<size_t>(sb.st_size);
            map_pos_ = 68 + static_cast<size_t>(hsize);
            madvise(m, map_size_, MADV_SEQUENTIAL);
        }
    }
    return true;
}

// ---- SLOW5 ASCII ------------------------------------------------------------------------------------------------------------
namespace {
// slow5_uint_check + strtoull (slow5_misc.c:103-120, 161-186): digits only, no leading zero on a longer number
bool ascii_uint(const char *p, size_t n, uint64_t max, uint64_t *out) {
    if (n == 0 || n > 20 || (n > 1 && p[0] == '0')) return false;
    unsigned __int128 v = 0;
    for (size_t i = 0; i < n; ++i) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + static_cast<unsigned>(p[i] - '0');
    }
    if (v > max) return false;
    *out = static_cast<uint64_t>(v);
    return true;
}
// slow5_strtod_check (slow5_misc.c:141-156, 360-376): digits, '.' and '-' only (no exponent), then strtod
bool ascii_double(const char *p, size_t n, double *out) {
    if (n == 0 || n >= 400) return false;
    char buf[400];
    for (size_t i = 0; i < n; ++i) {
        if (!((p[i] >= '0' && p[i] <= '9') || p[i] == '.' || p[i] == '-')) return false;
        buf[i] = p[i];
    }
    buf[n] = '\0';
    errno = 0;
    *out = strtod(buf, nullptr);
    return !(errno == ERANGE && (*out == HUGE_VAL || *out == -HUGE_VAL || *out == 0.0));
}
// slow5_ato_int16 (slow5_misc.c:122-139, 303-319): digits and '-' only, no leading zero on a longer token, strtol's value in range
bool ascii_int16(const char *p, size_t n, int16_t *out) {
    if (n == 0 || (n > 1 && p[0] == '0')) return false;
    size_t i = p[0] == '-' ? 1 : 0;
    long v = 0;
    bool plain = i < n && n - i <= 6;  // the common case: [-]digits
    for (size_t j = i; plain && j < n; ++j) {
        if (p[j] < '0' || p[j] > '9') plain = false;
        else v = v * 10 + (p[j] - '0');
    }
    if (!plain) {  // a '-' somewhere else, or a very long token: what strtol makes of it is what the reference takes
        if (n >= 64) return false;
        char buf[64];
        for (size_t j = 0; j < n; ++j) {
            if (!((p[j] >= '0' && p[j] <= '9') || p[j] == '-')) return false;
            buf[j] = p[j];
        }
        buf[n] = '\0';
        v = strtol(buf, nullptr, 10);
    } else if (i) {
        v = -v;
    }
    if (v > INT16_MAX || v < INT16_MIN) return false;
    *out = static_cast<int16_t>(v);
    return true;
}
}  // namespace

bool parse_slow5_line(const uint8_t *mem, size_t size, uint32_t n_aux, Blow5Record *rec, std::string *err) {
    rec->record_bytes = size;  // (the reference's -B accounting: the line without its newline, slow5.c:3214)
    const char *p = reinterpret_cast<const char *>(mem), *const end = p + size;
    const char *col[8];
    size_t len[8];
    int n_col = 0;
    while (n_col < 8) {
        const char *tab = static_cast<const char *>(memchr(p, '\t', static_cast<size_t>(end - p)));
        col[n_col] = p;
        len[n_col] = static_cast<size_t>((tab ? tab : end) - p);
        ++n_col;
        if (!tab) {
            p = nullptr;
            break;
        }
        p = tab + 1;
    }
    const auto bad = [&](const char *what) {
        *err = std::string("malformed SLOW5 record (") + what + (n_col > 0 ? ", read " + std::string(col[0], std::min<size_t>(len[0], 64)) : std::string()) + ")";
        return false;
    };
    if (n_col < 8) return bad("fewer than the eight primary columns");
    rec->read_id.assign(col[0], len[0]);
    uint64_t group = 0, n = 0;
    if (!ascii_uint(col[1], len[1], UINT32_MAX, &group)) return bad("read group");
    rec->read_group = static_cast<uint32_t>(group);
    if (!ascii_double(col[2], len[2], &rec->digitisation)) return bad("digitisation");
    if (!ascii_double(col[3], len[3], &rec->offset)) return bad("offset");
    if (!ascii_double(col[4], len[4], &rec->range)) return bad("range");
    if (!ascii_double(col[5], len[5], &rec->sampling_rate)) return bad("sampling rate");
    if (!ascii_uint(col[6], len[6], UINT64_MAX, &n)) return bad("raw signal length");
    if (n > len[7]) return bad("raw signal length exceeds the record");  // (every sample takes a character at least)
    rec->raw.resize(n);
    if (n > 0) {  // (a zero length leaves the signal column unread, slow5.c:2722-2725)
        const char *q = col[7], *const qend = q + len[7];
        uint64_t j = 0;
        for (;;) {
            const char *comma = static_cast<const char *>(memchr(q, ',', static_cast<size_t>(qend - q)));
            const char *tend = comma ? comma : qend;
            if (j >= n) return bad("more samples than the raw signal length");
            if (!ascii_int16(q, static_cast<size_t>(tend - q), &rec->raw[j])) return bad("raw signal");
            ++j;
            if (!comma) break;
            q = comma + 1;
        }
        if (j != n) return bad("fewer samples than the raw signal length");
    }
    // auxiliary columns: present in the record exactly when the header announced them (slow5.c:2778-2800); counted, not read
    uint32_t more = 0;
    for (; p; ++more) {
        const char *tab = static_cast<const char *>(memchr(p, '\t', static_cast<size_t>(end - p)));
        p = tab ? tab + 1 : nullptr;
    }
    if (more != n_aux) return bad(more < n_aux ? "auxiliary fields missing" : "auxiliary fields the header does not announce");
    return true;
}

bool Blow5Reader::open_ascii(const std::string &path) {
    ascii_ = true;
    eof_bytes_ = 0;
    record_press_ = signal_press_ = 0;
    rewind(fp_);
    {
        struct stat fsb;
        file_size_ = (fstat(fileno(fp_), &fsb) == 0 && S_ISREG(fsb.st_mode)) ? static_cast<uint64_t>(fsb.st_size) : UINT64_MAX;
    }
    char *line = nullptr;
    size_t cap = 0;
    uint64_t at = 0;
    const auto fail = [&](const std::string &what) {
        free(line);
        err_ = path + ": malformed SLOW5 header (" + what + ")";
        return false;
    };
    // every header line ends with a newline; returns the length without it, -1 at the end of the file
    const auto next_line = [&]() -> ssize_t {
        const ssize_t got = getline(&line, &cap, fp_);
        if (got <= 0 || line[got - 1] != '\n') return -1;
        at += static_cast<uint64_t>(got);
        line[got - 1] = '\0';
        return got - 1;
    };
    ssize_t n = next_line();
    unsigned ver[3];
    {  // "#slow5_version\tM.m.p", three uint8 (slow5.c:677-742); files newer than 1.0.0 are refused (slow5_defs.h: SLOW5_VERSION_ARRAY)
        if (n < 0 || strncmp(line, kAsciiFirst, sizeof kAsciiFirst - 1) != 0) return fail("no slow5_version line");
        const char *q = line + sizeof kAsciiFirst - 1;
        for (int k = 0; k < 3; ++k) {
            const char *dot = k < 2 ? strchr(q, '.') : q + strlen(q);
            uint64_t v;
            if (!dot || !ascii_uint(q, static_cast<size_t>(dot - q), 255, &v)) return fail("bad file version");
            ver[k] = static_cast<unsigned>(v);
            q = dot + 1;
        }
        if (ver[0] > 1 || (ver[0] == 1 && (ver[1] > 0 || ver[2] > 0))) return fail("file version newer than 1.0.0");
    }
    n = next_line();
    {
        uint64_t g;
        if (n < 0 || strncmp(line, kAsciiGroups, sizeof kAsciiGroups - 1) != 0) return fail("no num_read_groups line");
        const char *q = line + sizeof kAsciiGroups - 1;
        const char *tab = strchr(q, '\t');
        if (!ascii_uint(q, tab ? static_cast<size_t>(tab - q) : strlen(q), UINT32_MAX, &g) || g == 0) return fail("invalid number of read groups");
        n_groups_ = static_cast<uint32_t>(g);
    }
    // "@attr\tv0[\tv1...]" until the column types (slow5.c:1701-1760): the first value is read group 0's
    for (;;) {
        n = next_line();
        if (n < 0) return fail("no column types line");
        if (strncmp(line, kAsciiTypes, sizeof kAsciiTypes - 1) == 0) break;
        if (line[0] != '@') return fail("a line that is neither an attribute nor the column types");
        const char *tab = strchr(line, '\t');
        if (!tab) return fail("an attribute without a value");
        const char *vend = strchr(tab + 1, '\t');
        attrs_[std::string(line + 1, static_cast<size_t>(tab - line - 1))] = vend ? std::string(tab + 1, static_cast<size_t>(vend - tab - 1)) : std::string(tab + 1);
    }
    const auto extra_columns = [](const char *rest, uint32_t *count) {  // "" or "\tname[\tname...]"
        *count = 0;
        if (*rest == '\0') return true;
        if (*rest != '\t') return false;
        for (const char *q = rest; q; q = strchr(q + 1, '\t')) ++*count;
        return true;
    };
    uint32_t n_types = 0, n_names = 0;
    if (!extra_columns(line + sizeof kAsciiTypes - 1, &n_types)) return fail("column types");
    n = next_line();
    if (n < 0 || strncmp(line, kAsciiNames, sizeof kAsciiNames - 1) != 0 || !extra_columns(line + sizeof kAsciiNames - 1, &n_names))
        return fail("column names");
    if (n_types != n_names) return fail("as many auxiliary types as names expected");
    n_aux_ = n_types;
    free(line);
    data_begin_ = pos_ = at;
    limit_pos_ = limit_records_ = UINT64_MAX;
    struct stat sb;
    if (fstat(fileno(fp_), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
        void *m = mmap(nullptr, static_cast<size_t>(sb.st_size), PROT_READ, MAP_PRIVATE, fileno(fp_), 0);
        if (m != MAP_FAILED) {
            map_ = static_cast<const uint8_t *>(m);
            map_size_ = static_cast<size_t>(sb.st_size);
            map_pos_ = static_cast<size_t>(at);
            madvise(m, map_size_, MADV_SEQUENTIAL);
        }
    }
    return true;
}

int Blow5Reader::skip_one() {
    if (map_) {
        const uint8_t *m;
        size_t n;
        return next_view(&m, &n);
    }
    if (!fp_) return -1;
    if (ascii_) return next_mem(&buf_);
    uint64_t size = 0;
    const size_t got = fread(&size, 1, sizeof size, fp_);
    if (got != sizeof size) {
        if (got == sizeof kEof && memcmp(&size, kEof, sizeof kEof) == 0) return 0;
        err_ = "malformed BLOW5: missing end-of-file marker";
        return -1;
    }
    if (size > file_size_ || fseeko(fp_, static_cast<off_t>(size), SEEK_CUR) != 0) {
        err_ = "malformed BLOW5: truncated record";
        return -1;
    }
    pos_ += 8 + size;
    return 1;
}

bool Blow5Reader::select_records(uint64_t first, uint64_t count) {
    for (uint64_t i = 0; i < first; ++i) {
        const int rc = skip_one();
        if (rc < 0) return false;
        if (rc == 0) break;  // a range beyond the end of the file is empty, not an error
    }
    limit_records_ = count;
    return true;
}

bool Blow5Reader::select_shard(uint32_t r, uint32_t G) {
    if (G == 0 || r >= G) {
        err_ = "shard index out of range";
        return false;
    }
    if (file_size_ == UINT64_MAX || file_size_ < data_begin_ + eof_bytes_) {
        err_ = "sharding by byte ranges needs a regular, complete BLOW5 file";
        return false;
    }
    const uint64_t region = file_size_ - eof_bytes_ - data_begin_;
    const auto cut = [&](uint32_t k) { return data_begin_ + static_cast<uint64_t>(static_cast<unsigned __int128>(region) * k / G); };
    const uint64_t lo = cut(r);
    while (pos_ < lo) {
        const int rc = skip_one();
        if (rc < 0) return false;
        if (rc == 0) break;
    }
    limit_pos_ = r + 1 == G ? UINT64_MAX : cut(r + 1);  // the last shard runs into the end-of-file marker like an unsharded read
    return true;
}

int Blow5Reader::next_view(const uint8_t **mem, size_t *size) {
    if (!map_) return -2;  // caller falls back to next_mem()
    if (limit_records_ == 0 || pos_ >= limit_pos_) return 0;
    if (ascii_) {  // a record is a line (getline in slow5_get_next_mem, slow5.c:3200-3216); the file simply ends
        if (map_pos_ >= map_size_) return 0;
        const void *nl = memchr(map_ + map_pos_, '\n', map_size_ - map_pos_);
        if (!nl) {  // (slow5lib would drop the last character of such a line and carry on)
            err_ = "malformed SLOW5: the last record does not end with a newline (truncated file?)";
            return -1;
        }
        *mem = map_ + map_pos_;
        *size = static_cast<size_t>(static_cast<const uint8_t *>(nl) - *mem);
        map_pos_ += *size + 1;
        pos_ = map_pos_;
        if (limit_records_ != UINT64_MAX) --limit_records_;
        consumed_.store(map_pos_, std::memory_order_relaxed);
        return 1;
    }
    if (map_pos_ + sizeof kEof <= map_size_ && map_size_ - map_pos_ == sizeof kEof && memcmp(map_ + map_pos_, kEof, sizeof kEof) == 0) return 0;
    if (map_pos_ + 8 > map_size_) {
        err_ = "malformed BLOW5: missing end-of-file marker";
        return -1;
    }
    uint64_t sz;
    memcpy(&sz, map_ + map_pos_, 8);
    if (sz > map_size_ - map_pos_ - 8) {
        err_ = "malformed BLOW5: truncated record";
        return -1;
    }
    *mem = map_ + map_pos_ + 8;
    *size = static_cast<size_t>(sz);
    map_pos_ += 8 + static_cast<size_t>(sz);
    pos_ = map_pos_;
    if (limit_records_ != UINT64_MAX) --limit_records_;
    consumed_.store(map_pos_, std::memory_order_relaxed);
    return 1;
}

int Blow5Reader::next_mem(std::vector<uint8_t> *mem) {
    if (!fp_) return -1;
    if (limit_records_ == 0 || pos_ >= limit_pos_) return 0;
    if (ascii_) {
        char *line = nullptr;
        size_t cap = 0;
        const ssize_t got = getline(&line, &cap, fp_);
        if (got <= 0) {
            free(line);
            if (feof(fp_)) return 0;
            err_ = "reading the next SLOW5 record failed";
            return -1;
        }
        if (line[got - 1] != '\n') {
            free(line);
            err_ = "malformed SLOW5: the last record does not end with a newline (truncated file?)";
            return -1;
        }
        mem->assign(reinterpret_cast<const uint8_t *>(line), reinterpret_cast<const uint8_t *>(line) + got - 1);
        free(line);
        pos_ += static_cast<uint64_t>(got);
        if (limit_records_ != UINT64_MAX) --limit_records_;
        return 1;
    }
    uint64_t size = 0;
    const size_t got = fread(&size, 1, sizeof size, fp_);
    if (got != sizeof size) {
        if (got == sizeof kEof && memcmp(&size, kEof, sizeof kEof) == 0) return 0;
        err_ = "malformed BLOW5: missing end-of-file marker";
        return -1;
    }
    if (size > file_size_ || (file_size_ == UINT64_MAX && size > (uint64_t(1) << 32))) {  // not a plausible record size
        err_ = "malformed BLOW5: truncated record";
        return -1;
    }
    mem->resize(size);
    if (size && fread(mem->data(), 1, size, fp_) != size) {
        err_ = "malformed BLOW5: truncated record";
        return -1;
    }
    pos_ += 8 + size;
    if (limit_records_ != UINT64_MAX) --limit_records_;
    return 1;
}

bool Blow5Reader::parse(const std::vector<uint8_t> &mem, Blow5Record *rec, std::string *err) const {
    return parse(mem.data(), mem.size(), rec, err);
}

bool Blow5Reader::parse(const uint8_t *mem, size_t size, Blow5Record *rec, std::string *err) const {
    if (ascii_) return parse_slow5_line(mem, size, n_aux_, rec, err);
    return parse_blow5_record(mem, size, record_press_ == 1, signal_press_ == 1, rec, err);
}

namespace {
// the fields and the signal of one record from its (inflated) payload [p, end)
bool parse_payload(const uint8_t *p, const uint8_t *end, int signal_svb, Blow5Record *rec, std::string *err);
}  // namespace

bool parse_blow5_record(const uint8_t *mem, size_t size, int record_zlib, int signal_svb, Blow5Record *rec, std::string *err) {
    rec->record_bytes = size;
    const uint8_t *p = mem, *end = p + size;
    if (record_zlib) {
        const uint8_t *inflated = nullptr;  // this thread's buffer, valid until its next record
        size_t inflated_len = 0;
        if (!inflate_all(mem, size, &inflated, &inflated_len)) {
            *err = "malformed BLOW5: record does not inflate";
            return false;
        }
        p = inflated;
        end = p + inflated_len;
    }
    return parse_payload(p, end, signal_svb, rec, err);
}

// Two records by one thread: their zlib streams are inflated side by side (inflate.hpp: two dependence chains keep a core busier
// than one); everything else is parse_blow5_record's, and so is the result -- a record the paired decoder declines goes through
// the single-record path (own decoder, then zlib), errors are reported per record.
void parse_blow5_record_pair(const uint8_t *const mem[2], const size_t size[2], int record_zlib, int signal_svb, Blow5Record *const rec[2],
                             std::string *const err[2], bool ok[2]) {
    if (!record_zlib) {
        for (int k = 0; k < 2; ++k) ok[k] = parse_blow5_record(mem[k], size[k], record_zlib, signal_svb, rec[k], err[k]);
        return;
    }
    thread_local std::vector<uint8_t> buf0, buf1;
    std::vector<uint8_t> *const bufs[2] = {&buf0, &buf1};
    size_t len[2] = {0, 0};
    bool inflated[2];
    fast_inflate_zlib_pair(mem, size, bufs, len, inflated);
    for (int k = 0; k < 2; ++k) {
        if (!inflated[k]) {  // declined: the single-record path decides (zlib's own inflate as the last word)
            ok[k] = parse_blow5_record(mem[k], size[k], record_zlib, signal_svb, rec[k], err[k]);
            continue;
        }
        rec[k]->record_bytes = size[k];
        ok[k] = parse_payload(bufs[k]->data(), bufs[k]->data() + len[k], signal_svb, rec[k], err[k]);
    }
}

namespace {
bool parse_payload(const uint8_t *p, const uint8_t *end, int signal_svb, Blow5Record *rec, std::string *err) {
    uint16_t idlen;
    uint64_t len;
    bool ok = take(p, end, &idlen) && static_cast<size_t>(end - p) >= idlen;
    if (ok) {
        rec->read_id.assign(reinterpret_cast<const char *>(p), idlen);
        p += idlen;
        ok = take(p, end, &rec->read_group) && take(p, end, &rec->digitisation) && take(p, end, &rec->offset) &&
             take(p, end, &rec->range) && take(p, end, &rec->sampling_rate) && take(p, end, &len);
    }
    if (ok) {
        if (!signal_svb) {
            ok = len <= static_cast<uint64_t>(end - p) / 2;  // (not len * 2: a corrupt length must not wrap around)
            if (ok) {
                rec->raw.resize(len);
                memcpy(rec->raw.data(), p, len * 2);
            }
        } else {
            ok = static_cast<uint64_t>(end - p) >= len && decode_svb_zd(p, len, &rec->raw);
        }
    }
    if (!ok) *err = "malformed BLOW5 record (read " + rec->read_id + ")";
    return ok;
}
}  // namespace

void Blow5Reader::parse_pair(const uint8_t *const mem[2], const size_t size[2], Blow5Record *const rec[2], std::string *const err[2], bool ok[2]) const {
    if (ascii_) {
        for (int k = 0; k < 2; ++k) ok[k] = parse_slow5_line(mem[k], size[k], n_aux_, rec[k], err[k]);
        return;
    }
    parse_blow5_record_pair(mem, size, record_press_ == 1, signal_press_ == 1, rec, err, ok);
}

int Blow5Reader::next(Blow5Record *rec) {
    if (map_) {  // one position per reader: the iterator that a selection walked is the one that reads on
        const uint8_t *mem;
        size_t size;
        const int rc = next_view(&mem, &size);
        if (rc <= 0) return rc;
        return parse(mem, size, rec, &err_) ? 1 : -1;
    }
    const int rc = next_mem(&buf_);
    if (rc <= 0) return rc;
    return parse(buf_, rec, &err_) ? 1 : -1;
}

}  // namespace sfa
