#include "refio.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace sfa {

bool read_fasta(const std::string &path, std::vector<FastaRecord> *out, std::string *err) {
    gzFile fp = gzopen(path.c_str(), "r");
    if (!fp) {
        *err = "cannot open " + path;
        return false;
    }
    out->clear();
    std::vector<char> buf(1 << 16);
    std::string line;
    bool in_seq = false, in_qual = false;
    size_t qual_seen = 0;
    auto handle = [&](const std::string &l) {
        if (l.empty()) return;
        if (!in_qual && (l[0] == '>' || l[0] == '@')) {
            FastaRecord r;
            size_t e = 1;
            while (e < l.size() && !isspace(static_cast<unsigned char>(l[e]))) ++e;
            r.name = l.substr(1, e - 1);
            out->push_back(r);
            in_seq = true;
        } else if (in_seq && l[0] == '+') {
            in_seq = false;
            in_qual = true;  // FASTQ quality block: skipped until as many characters as the sequence were seen
        } else if (in_seq && !out->empty()) {
            for (char c : l)
                if (isgraph(static_cast<unsigned char>(c))) out->back().seq.push_back(c);
        } else if (in_qual) {
            qual_seen += l.size();
            if (qual_seen >= out->back().seq.size()) {
                in_qual = false;
                qual_seen = 0;
            }
        }
    };
    int n;
    while ((n = gzread(fp, buf.data(), static_cast<unsigned>(buf.size()))) > 0) {
        for (int i = 0; i < n; ++i) {
            const char c = buf[i];
            if (c == '\n') {
                if (!line.empty() && line.back() == '\r') line.pop_back();
                handle(line);
                line.clear();
            } else {
                line.push_back(c);
            }
        }
    }
    if (!line.empty()) handle(line);
    gzclose(fp);
    if (n < 0) {
        *err = "error while reading " + path;
        return false;
    }
    return true;
}

bool read_kmer_model(const std::string &path, std::vector<float> *level_mean, uint32_t *k_out, std::string *err) {
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) {
        *err = "cannot open k-mer model " + path;
        return false;
    }
    uint32_t k = 9;  // MAX_KMER_SIZE unless a "#k" line says otherwise (src/model.c:40-41)
    size_t want = static_cast<size_t>(1) << (2 * k);
    level_mean->clear();
    char *line = nullptr;
    size_t cap = 0;
    ssize_t len;
    int line_no = 0;
    bool ok = true;
    while ((len = getline(&line, &cap, fp)) != -1) {
        ++line_no;
        if (line[0] == '#' || line[0] == '\n' || line[0] == '\r' || strncmp(line, "kmer\t", 5) == 0) {
            char key[64];
            int val = 0;
            if (sscanf(line, "%63s\t%d", key, &val) == 2 && strcmp(key, "#k") == 0) {
                if (val <= 0 || val > 9) {
                    *err = "k-mer size (#k " + std::to_string(val) + ") in " + path + " is invalid (1..9)";
                    ok = false;
                    break;
                }
                k = static_cast<uint32_t>(val);
                want = static_cast<size_t>(1) << (2 * k);
            }
            continue;
        }
        char kmer[64];
        float mean, stdv;
        if (sscanf(line, "%63s\t%f\t%f", kmer, &mean, &stdv) != 3) {
            *err = path + " is corrupted at line " + std::to_string(line_no);
            ok = false;
            break;
        }
        level_mean->push_back(mean);
        if (level_mean->size() > want) {
            *err = path + " has too many entries (expected " + std::to_string(want) + " k-mers)";
            ok = false;
            break;
        }
    }
    free(line);
    fclose(fp);
    if (ok && level_mean->size() != want) {
        *err = path + " ended prematurely: expected " + std::to_string(want) + " k-mers, found " + std::to_string(level_mean->size());
        ok = false;
    }
    *k_out = k;
    return ok;
}

}  // namespace sfa
