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

// read_model(), src/model.c:38-131: text table "kmer<TAB>level_mean<TAB>level_stdv[...]" with optional "#k<TAB>K" lines.
// Same acceptance as the reference (tests/test_kmer_model_reader.py holds the table of cases): lines beginning with '#', empty
// lines and exactly three header spellings are skipped; every other line is a table row and counts, whether it parses or not
// (a row without three fields is reported in *warnings and the run goes on, as the reference only logs it); without a #k line
// k = 9; a table longer or shorter than 4^k entries is an error, and so is #k outside 1..9.
bool read_kmer_model(const std::string &path, std::vector<float> *level_mean, uint32_t *k, std::string *err, std::string *warnings = nullptr);

}  // namespace sfa
