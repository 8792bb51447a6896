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

bool read_kmer_model(const std::string &path, std::vector<float> *level_mean, uint32_t *k_out, std::string *err, std::string *warnings) {
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
    int line_no = 0;
    bool ok = true;
    // the three header spellings the reference knows, newline included (src/model.c:63-65); any other line that begins with
    // "kmer" is a table row to it -- one that does not parse, is counted, and makes the table one entry too long
    static const char *const kHeaders[3] = {"kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\tweight\n", "kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n",
                                            "kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\tig_lambda\tweight\n"};
    while (getline(&line, &cap, fp) != -1) {
        ++line_no;
        if (line[0] == '#' || line[0] == '\n' || line[0] == '\r' || !strcmp(line, kHeaders[0]) || !strcmp(line, kHeaders[1]) || !strcmp(line, kHeaders[2])) {
            char key[1000];
            int val = 0;
            if (sscanf(line, "%999s\t%d", key, &val) == 2 && strcmp(key, "#k") == 0) {  // src/model.c:69-84; may come more than once, the last one counts
                if (val <= 0) {
                    *err = "k-mer size (#k\t" + std::to_string(val) + ") in file " + path + " is invalid.";
                    ok = false;
                    break;
                }
                if (val > 9) {
                    *err = "k-mer size (#k\t" + std::to_string(val) + ") in file " + path + " larger than MAX_KMER_SIZE (9).";
                    ok = false;
                    break;
                }
                k = static_cast<uint32_t>(val);
                want = static_cast<size_t>(1) << (2 * k);
            }
            continue;
        }
        // src/model.c:90-100: a row that does not give three fields is reported and COUNTED, the run goes on.  Its level_mean is
        // the parsed one when the row got that far; otherwise the reference leaves its table entry as malloc returned it -- 0 here.
        char kmer[64];
        float mean = 0.f, stdv = 0.f;
        const int got = sscanf(line, "%63s\t%f\t%f", kmer, &mean, &stdv);
        if (got != 3 && warnings)
            *warnings += "File " + path + " is corrupted at line " + std::to_string(line_no) + ". Does the format adhere to examples at test/r9-models?\n";
        level_mean->push_back(got >= 2 ? mean : 0.f);
        if (level_mean->size() > want) {  // src/model.c:101-106
            *err = "File " + path + " has too many entries. Expected " + std::to_string(want) + " kmers in the model, but file had more than that";
            ok = false;
            break;
        }
    }
    free(line);
    fclose(fp);
    if (ok && level_mean->size() != want) {  // src/model.c:111-116
        *err = "File " + path + " prematurely ended. Expected " + std::to_string(want) + " kmers in the model, but file had only " +
               std::to_string(level_mean->size());
        ok = false;
    }
    *k_out = k;
    return ok;
}

}  // namespace sfa
