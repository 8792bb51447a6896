// host_asan.cpp -- the host-side units under AddressSanitizer / UBSan (CPU only; the GPU pool forbids GPU sanitizers):
// BLOW5 reader incl. the own inflate and the SSSE3 StreamVByte decoder, event detection, query selection incl. the RNA
// adaptor / poly-A search, corrupted files, planner.  Built and run by tests/test_c_host.py.
//   usage: host_asan <scratch dir> <blow5>...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/sigfish_amd.h"

extern "C" void sfa_set_error_(const char *) {}  // defined in sfa_capi.hip, which is not part of this build

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
        while ((rc = sfa_blow5_next(f, &id, meta, &raw, &n)) == 1) {
        }
        if (rc < 0) ++rejected;
        sfa_blow5_close(f);
    }
    std::vector<int64_t> qo(5001, 0);
    for (int i = 0; i < 5000; ++i) qo[i + 1] = qo[i] + (rng() % 9 == 0 ? rng() % 2049 : 250);
    const int32_t jl[2] = {29898, 29898};
    std::vector<int32_t> slot(5000);
    sfa_plan_info_t info;
    for (int w : {0, 1, 2, 4})
        if (sfa_plan_batch(qo.data(), 5000, jl, 2, 0, 0, w, slot.data(), &info) != 0) return 7;
    printf("%ld reads, %ld events, %d of 300 corrupt files rejected, %d quads\n", reads, events, rejected, info.n_quads);
    return (reads > 0 && rejected > 250) ? 0 : 8;
}
