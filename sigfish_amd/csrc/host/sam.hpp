// sam.hpp -- SAM output of the `dtw` path: the winner's warp path, the reference-to-event map
// and the signal-space CIGAR-like "ss" string.
//
// The GPU returns the winning alignment's end and start columns; the path itself is recovered on the host by
// re-filling only the band [start,end] x qlen and walking it back with the reference's rule.  This is exact: every
// cell ON the optimal path has the same cost in the band as in the full matrix (its cost is the prefix sum along
// the path, which lies inside the band), cells off the path can only get more expensive in the band, and the
// traceback prefers diagonal > left > up among neighbours EQUAL to the minimum -- so the same predecessor wins.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/sigfish_amd.h"

namespace sfa {

struct WarpPath {
    std::vector<int32_t> px, py;  // query row, reference column (strand-array coordinates), start -> end
};

// query: z-normalised events in DP order (already reversed for RNA); y: the (contig,strand) array, rlen columns;
// [col_st, col_end]: alignment columns in that array.  std_dtw: --dtw-std recurrence (row 0 cumulative from column 0).
WarpPath band_traceback(const float *query, int32_t qlen, const float *y, int32_t rlen, int32_t col_st, int32_t col_end, bool std_dtw);

// path_to_map(), src/sigfish.c:530-571: per reference column of the alignment the first / last query row mapped to it
// (-1 / -1: a column the path crosses without advancing in the query); pairs[2*i], pairs[2*i+1] = start, stop of column
// pos_st + i (layout of index_pair_t, src/sigfish.h:141-144).  Length = last column - first column + 1.
std::vector<int32_t> path_to_pairs(const WarpPath &path);

// path_to_map + r2qevent_map_to_ss + sam_str (src/sigfish.c:530-571, 663-794) for one read
std::string sam_record(const sfa_result_t &row, const WarpPath &path, const char *read_id, const char *rname, const sfa_event_t *events,
                       int64_t qstart, int64_t qend, bool rna);

}  // namespace sfa
