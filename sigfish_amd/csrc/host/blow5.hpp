// blow5.hpp -- minimal sequential BLOW5 reader: just what the `dtw` path consumes.
//
// On-disk layout (slow5 spec 0.2.0; reference reader: slow5lib/src/slow5.c:792-835 header, 3218-3266 record
// framing, 2806-2925 primary fields; slow5lib/src/slow5_press.c:76-146 method codes, 1085-1135 svb-zd):
//   magic "BLOW5\1" | version u8[3] | record_press u8 | num_read_groups u32 | signal_press u8 (>= 0.2.0) |
//   zero pad to byte 64 | ascii_header_size u32 | ascii header ("@attr\tv0[\tv1..]" lines, "#types", "#names") |
//   records: [u64 size][payload] ... | EOF marker "5WOLB"
//   payload (after inflating the whole record when record_press == zlib):
//   u16 id_len, id, u32 read_group, f64 digitisation, f64 offset, f64 range, f64 sampling_rate, u64 len,
//   signal bytes, auxiliary fields (ignored here).  With signal_press == svb-zd, `len` is the compressed byte
//   count and the signal is u32 n + StreamVByte(keys ceil(n/4) B, data 1-4 B little endian) of zig-zag deltas.
//
// ... and its text twin, SLOW5 ASCII (slow5_open takes either, by the file's extension: slow5lib/src/slow5.c:4219-4229; here
// by content).  Layout (slow5.c:661-790 header, 1701-1990 attribute / type / name lines, 3200-3216 + 2643-2790 records):
//   "#slow5_version\tM.m.p" | "#num_read_groups\tN" | "@attr\tv0[\tv1..]" lines | "#char*\tuint32_t\tdouble\tdouble\tdouble\t
//   double\tuint64_t\tint16_t*[\taux types]" | "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\t
//   len_raw_signal\traw_signal[\taux names]" | one line per record: the eight primary columns, tab separated, the signal as
//   comma separated integers, then the auxiliary columns (counted, not interpreted).  No compression, no end-of-file marker.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace sfa {

struct Blow5Record {
    std::string read_id;
    uint32_t read_group = 0;
    double digitisation = 0, offset = 0, range = 0, sampling_rate = 0;
    std::vector<int16_t> raw;
    uint64_t record_bytes = 0;  // on-disk payload size (the reference's -B accounting, sigfish.c:304)
};

// One record's bytes -> fields + samples, given the file's compression methods (what Blow5Reader::parse does; free-standing
// for callers that hold record bytes without a reader, e.g. the fallback of the device-side decoder)
bool parse_blow5_record(const uint8_t *mem, size_t size, int record_zlib, int signal_svb, Blow5Record *rec, std::string *err);
// One line of a SLOW5 ASCII file (without its newline) -> fields + samples, with slow5lib's acceptance rules for the numbers
// (slow5_misc.c:103-156: no signs or leading zeros on the unsigned columns, digits '.' '-' only in the doubles);
// `n_aux`: auxiliary columns the header announced -- the line must carry exactly that many more
bool parse_slow5_line(const uint8_t *mem, size_t size, uint32_t n_aux, Blow5Record *rec, std::string *err);
// ... and two of them by one thread, their zlib streams inflated side by side (same results, record by record)
void parse_blow5_record_pair(const uint8_t *const mem[2], const size_t size[2], int record_zlib, int signal_svb, Blow5Record *const rec[2],
                             std::string *const err[2], bool ok[2]);

class Blow5Reader {
  public:
    ~Blow5Reader() { close(); }
    // returns false and sets error() on failure
    bool open(const std::string &path);
    void close();
    // 1: record read, 0: clean end of file (EOF marker seen), -1: error
    int next(Blow5Record *rec);
    // the two halves of next(), so that decompression can run on worker threads (the reference does the same:
    // slow5_get_next_mem in load_db, slow5_rec_depress_parse in parse_single, src/sigfish.c:289,322)
    int next_mem(std::vector<uint8_t> *mem);
    bool parse(const std::vector<uint8_t> &mem, Blow5Record *rec, std::string *err) const;
    // zero-copy variant: the file is mapped, a record is a (pointer, size) view into the mapping (valid until close())
    int next_view(const uint8_t **mem, size_t *size);
    bool parse(const uint8_t *mem, size_t size, Blow5Record *rec, std::string *err) const;
    void parse_pair(const uint8_t *const mem[2], const size_t size[2], Blow5Record *const rec[2], std::string *const err[2], bool ok[2]) const;
    // first value (read group 0) of a header attribute, or nullptr (slow5_hdr_get(attr, 0, hdr))
    const char *attr(const std::string &key) const;
    uint32_t num_read_groups() const { return n_groups_; }
    bool mapped() const { return map_ != nullptr; }           // next_view() works (regular file, mmap succeeded)
    bool ascii() const { return ascii_; }                     // SLOW5 ASCII: records are lines (host parsing only)
    bool records_zlib() const { return record_press_ == 1; }  // the whole record is a zlib stream
    bool signal_svb() const { return signal_press_ == 1; }    // the signal is StreamVByte of zig-zag deltas
    const std::string &error() const { return err_; }
    // Walk the mapping ahead of next_view() on a helper thread, one byte per page, at most `window` bytes ahead: the first
    // touch of a mapped page is a fault (0.25-0.4 us; a 4 KB record is a page), and next_view() runs on the caller's one
    // thread -- 0.16 s of "loading" per 400 000 records that nothing overlapped.  The helper takes the faults instead.
    void start_prefault(size_t window = size_t(256) << 20);
    // One rank's part of a read-sharded run.  Records are framed by their u64 size prefixes and nothing else
    // (slow5lib/src/slow5.c:3218-3266), so skipping means walking the prefixes: one touch per record, no decompression.
    //   select_records(first, count)  records [first, first + count) by position in the file (count = UINT64_MAX: to the end)
    //   select_shard(r, G)            the records whose size prefix starts in the r-th of G equal byte slices of the record
    //                                 region: needs no record count beforehand, balances the ranks by bytes, and the G shards
    //                                 together are every record exactly once, in file order (regular files only)
    // Call after open(), before the first next*().  Past the selection next*() return 0, as at the end of the file.
    bool select_records(uint64_t first, uint64_t count);
    bool select_shard(uint32_t r, uint32_t G);

  private:
    void stop_prefault();
    bool open_ascii(const std::string &path);
    bool ascii_ = false;
    uint32_t n_aux_ = 0;                   // auxiliary columns of an ASCII file
    size_t eof_bytes_ = 5;                 // the end-of-file marker behind the records (none in ASCII)
    int skip_one();                        // 1: a record skipped, 0: end of file, -1: error
    uint64_t data_begin_ = 0;              // offset of the first record's size prefix
    uint64_t pos_ = 0;                     // offset of the next record's size prefix (both iterators keep it)
    uint64_t limit_pos_ = UINT64_MAX;      // selection: records starting at or beyond this offset are not ours
    uint64_t limit_records_ = UINT64_MAX;  // selection: records still to hand out
    std::thread prefault_;
    std::atomic<bool> prefault_quit_{false};
    std::atomic<size_t> consumed_{0};  // copy of map_pos_ for the helper
    FILE *fp_ = nullptr;
    const uint8_t *map_ = nullptr;  // whole file, read-only mapping (nullptr: mmap unavailable, fall back to fread)
    size_t map_size_ = 0, map_pos_ = 0;
    uint64_t file_size_ = UINT64_MAX;  // upper bound for every size field read from the file (UINT64_MAX: not a regular file)
    uint8_t record_press_ = 0, signal_press_ = 0;
    uint32_t n_groups_ = 1;
    std::map<std::string, std::string> attrs_;
    std::string err_;
    std::vector<uint8_t> buf_, inflated_;
};

}  // namespace sfa
