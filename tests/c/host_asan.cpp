// host_asan.cpp -- the host-side units under AddressSanitizer / UBSan (CPU only; the GPU pool forbids GPU sanitizers):
// BLOW5 / SLOW5 ASCII reader incl. the own inflate and the SSSE3 StreamVByte decoder, event detection, query selection incl. the RNA
// adaptor / poly-A search, corrupted files, planner.  Built and run by tests/test_c_host.py.
//   usage: host_asan <scratch dir> <blow5>...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/sigfish_amd.h"

extern "C" void sfa_set_error_(const char *) {}  // defined in sfa_context.hip, which is not part of this build

static std::vector<unsigned char> slurp(const char *p) {
    std::vector<unsigned char> b;
    FILE *f = fopen(p, "rb");
    if (!f) return b;
    fseek(f, 0, SEEK_END);
    b.resize(static_cast<size_t>(ftell(f)));
    fseek(f, 0, SEEK_SET);
    if (fread(b.data(), 1, b.size(), f) != b.size()) b.clear();
    fclose(f);
    return b;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string scratch = std::string(argv[1]) + "/fz.blow5";
    long reads = 0, events = 0;
    for (int a = 2; a < argc; ++a) {
        const char *path = argv[a];
        sfa_blow5_t *f = sfa_blow5_open(path);
        if (!f) return 3;
        const bool rna = strstr(path, "rna") != nullptr;
        const char *id;
        double meta[4];
        const int16_t *raw;
        int64_t n;
        while (sfa_blow5_next(f, &id, meta, &raw, &n) == 1) {
            std::vector<sfa_event_t> ev(static_cast<size_t>(n) + 2);
            const int64_t ne = sfa_detect_events(raw, n, meta[0], meta[1], meta[2], rna ? 1 : 0, ev.data(), static_cast<int64_t>(ev.size()));
            if (ne < 0) return 4;
            for (int prefix : {50, 0, -1}) {
                if (prefix < 0 && !rna) continue;
                std::vector<sfa_event_t> e2(ev.begin(), ev.begin() + ne);
                int64_t qs = 0, qe = 0;
                sfa_select_query(e2.data(), ne, raw, n, meta[0], meta[1], meta[2], prefix, 250, rna ? SFA_RNA : 0, 0, &qs, &qe);
            }
            ++reads;
            events += ne;
        }
        sfa_blow5_close(f);
    }
    std::mt19937 rng(7);
    const std::vector<unsigned char> src = slurp(argv[2]);
    if (src.size() < 200) return 5;
    int rejected = 0;
    for (int it = 0; it < 300; ++it) {
        std::vector<unsigned char> b = src;
        if (it % 3 == 0) {
            for (int k = 0; k < 4; ++k) b[rng() % b.size()] = static_cast<unsigned char>(rng());
        } else if (it % 3 == 1) {
            b.resize(rng() % b.size());
        } else {
            const size_t p = 64 + rng() % (b.size() - 80);
            for (int k = 0; k < 8; ++k) b[p + k] = static_cast<unsigned char>(rng());
        }
        FILE *o = fopen(scratch.c_str(), "wb");
        if (!o) return 6;
        fwrite(b.data(), 1, b.size(), o);
        fclose(o);
        sfa_blow5_t *f = sfa_blow5_open(scratch.c_str());
        if (!f) {
            ++rejected;
            continue;
        }
        const char *id;
        double meta[4];
        const int16_t *raw;
        int64_t n;
        int rc;
        // a third of the damaged files are read as one rank of a sharded run would: the walk over the size prefixes to the
        // shard's / range's first record must stay inside the file whatever the prefixes say
        if (it % 3 == 1) (void)sfa_blow5_select_shard(f, static_cast<int32_t>(rng() % 4), 4);
        if (it % 3 == 2) (void)sfa_blow5_select_records(f, static_cast<int64_t>(rng() % 6), (rng() & 1) ? -1 : static_cast<int64_t>(rng() % 4));
        while ((rc = sfa_blow5_next(f, &id, meta, &raw, &n)) == 1) {
        }
        if (rc < 0) ++rejected;
        sfa_blow5_close(f);
    }
    // the shards of an intact file: every record exactly once (counts; contents are compared in tests/test_blow5_shards.py)
    for (int a = 2; a < argc; ++a)
        for (int G : {1, 2, 3, 7, 64}) {
            long got = 0, whole = 0;
            for (int r = -1; r < G; ++r) {
                sfa_blow5_t *f = sfa_blow5_open(argv[a]);
                if (!f) return 12;
                if (r >= 0 && sfa_blow5_select_shard(f, r, G) != 0) return 13;
                const char *id;
                double meta[4];
                const int16_t *raw;
                int64_t n;
                while (sfa_blow5_next(f, &id, meta, &raw, &n) == 1) ++(r < 0 ? whole : got);
                sfa_blow5_close(f);
            }
            if (got != whole || whole == 0) return 14;
        }
    // SLOW5 ASCII, the text twin: the first file's records written out as text, read back, then damaged 400 times (bytes
    // replaced by separators, digits, signs and noise; truncation; lines glued together), whole and as shards / ranges
    int text_rejected = 0;
    {
        const std::string text_path = std::string(argv[1]) + "/fz.slow5";
        std::string text = "#slow5_version\t0.2.0\n#num_read_groups\t1\n@experiment_type\tgenomic_dna\n"
                           "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\tuint8_t\n"
                           "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\tflag\n";
        long written = 0, samples = 0;
        {
            sfa_blow5_t *f = sfa_blow5_open(argv[2]);
            if (!f) return 15;
            const char *id;
            double meta[4];
            const int16_t *raw;
            int64_t n;
            char num[64];
            while (sfa_blow5_next(f, &id, meta, &raw, &n) == 1) {
                text += std::string(id) + "\t0";
                for (int k = 0; k < 4; ++k) {
                    snprintf(num, sizeof num, "\t%.10f", meta[k]);
                    text += num;
                }
                text += "\t" + std::to_string(n) + "\t";
                for (int64_t i = 0; i < n; ++i) text += (i ? "," : "") + std::to_string(raw[i]);
                text += "\t1\n";
                ++written;
                samples += n;
            }
            sfa_blow5_close(f);
        }
        const auto read_all = [&](const std::string &body, int mode, long *got, long *got_samples) {
            FILE *o = fopen(text_path.c_str(), "wb");
            if (!o) return -2;
            fwrite(body.data(), 1, body.size(), o);
            fclose(o);
            sfa_blow5_t *f = sfa_blow5_open(text_path.c_str());
            if (!f) return -1;
            if (mode == 1) (void)sfa_blow5_select_shard(f, static_cast<int32_t>(rng() % 3), 3);
            if (mode == 2) (void)sfa_blow5_select_records(f, static_cast<int64_t>(rng() % 4), (rng() & 1) ? -1 : static_cast<int64_t>(rng() % 3));
            const char *id;
            double meta[4];
            const int16_t *raw;
            int64_t n;
            int rc;
            while ((rc = sfa_blow5_next(f, &id, meta, &raw, &n)) == 1) {
                ++*got;
                for (int64_t i = 0; i < n; ++i) *got_samples += raw[i] != INT16_MIN;  // (touch every sample)
            }
            sfa_blow5_close(f);
            return rc;
        };
        long got = 0, got_samples = 0;
        if (read_all(text, 0, &got, &got_samples) != 0 || got != written || got_samples != samples || written == 0) return 16;
        static const char kNoise[] = "\t\t,,\n\n--..0123456789ex#@ \0";
        for (int it = 0; it < 400; ++it) {
            std::string b = text;
            switch (it % 4) {
                case 0: for (int k = 0; k < 3; ++k) b[rng() % b.size()] = kNoise[rng() % (sizeof kNoise - 1)]; break;
                case 1: b.resize(rng() % b.size()); break;
                case 2: b.erase(rng() % b.size(), 1 + rng() % 40); break;
                default: b.insert(rng() % b.size(), std::string(1 + rng() % 5, kNoise[rng() % (sizeof kNoise - 1)])); break;
            }
            long g = 0, gs = 0;
            if (read_all(b, it % 3, &g, &gs) < 0) ++text_rejected;
        }
        if (text_rejected < 150) return 17;
    }
    // the reader's threads inflate records two at a time (sfa_inflate_zlib_pair): pairs of the file's own records, one or both
    // damaged (flipped bits, truncation, random tails), into generous and into too-small buffers
    int pair_ok = 0, pair_declined = 0;
    {
        std::vector<std::vector<unsigned char>> recs;
        uint32_t hsize;
        memcpy(&hsize, src.data() + 64, 4);
        size_t p = 68 + hsize;
        while (p + 8 <= src.size() && memcmp(src.data() + p, "5WOLB", 5) != 0) {
            uint64_t sz;
            memcpy(&sz, src.data() + p, 8);
            if (sz > src.size() - p - 8) break;
            recs.emplace_back(src.begin() + p + 8, src.begin() + p + 8 + sz);
            p += 8 + sz;
        }
        if (recs.size() < 2) return 9;
        std::vector<unsigned char> o0(1 << 16), o1(1 << 16);
        for (int it = 0; it < 3000; ++it) {
            std::vector<unsigned char> a = recs[rng() % recs.size()], b = recs[rng() % recs.size()];
            for (std::vector<unsigned char> *v : {&a, &b}) {
                switch (rng() % 5) {
                    case 0: (*v)[rng() % v->size()] ^= static_cast<unsigned char>(1u << (rng() % 8)); break;
                    case 1: v->resize(rng() % v->size() + 1); break;
                    case 2: for (int k = 0; k < 9; ++k) v->push_back(static_cast<unsigned char>(rng())); break;
                    default: break;  // intact
                }
            }
            const size_t c0 = (it % 7 == 0) ? 100 : o0.size(), c1 = (it % 11 == 0) ? 0 : o1.size();
            int64_t len[2];
            if (sfa_inflate_zlib_pair(a.data(), a.size(), b.data(), b.size(), o0.data(), c0, o1.data(), c1, len) != 0) return 10;
            for (int k = 0; k < 2; ++k) (len[k] >= 0 ? pair_ok : pair_declined) += 1;
        }
        if (pair_ok < 2000 || pair_declined < 500) return 11;
    }
    std::vector<int64_t> qo(5001, 0);
    for (int i = 0; i < 5000; ++i) qo[i + 1] = qo[i] + (rng() % 9 == 0 ? rng() % 2049 : 250);
    const int32_t jl[2] = {29898, 29898};
    std::vector<int32_t> slot(5000);
    sfa_plan_info_t info;
    for (int w : {0, 1, 2, 4})
        if (sfa_plan_batch(qo.data(), 5000, jl, 2, 0, 0, w, slot.data(), &info) != 0) return 7;
    printf("%ld reads, %ld events, %d of 300 corrupt files rejected, %d of 400 corrupt text files, %d / %d paired streams inflated / declined, %d quads\n",
           reads, events, rejected, text_rejected, pair_ok, pair_declined, info.n_quads);
    return (reads > 0 && rejected > 200) ? 0 : 8;  // (a damaged file whose shard is empty or lies in front of the damage reads clean)
}
