// refio.hpp -- FASTA(.gz) and k-mer model file readers for the `dtw` host.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sfa {

struct FastaRecord {
    std::string name, seq;
};

// kseq.h semantics (src/kseq.h via src/genref.c:88-94): '>' header, name = first word, sequence lines
// concatenated with whitespace dropped; FASTQ records ('@' header, '+' separator) are accepted as well.
bool read_fasta(const std::string &path, std::vector<FastaRecord> *out, std::string *err);

// read_model(), src/model.c:38-131: text table "kmer<TAB>level_mean<TAB>level_stdv[...]" with optional
// "#k<TAB>K" line, comment/header lines skipped.  Without a #k line the reference assumes k = 9.
bool read_kmer_model(const std::string &path, std::vector<float> *level_mean, uint32_t *k, std::string *err);

}  // namespace sfa
