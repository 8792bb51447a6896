// events.hpp -- host pre-DP stages of the `dtw` path: raw signal -> pA -> events ->
// query window -> z-normalised query.  Arithmetic follows the reference operation by operation (types and
// evaluation order included) so that the event means handed to the GPU are bit-identical.
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/sigfish_amd.h"

namespace sfa {

// event_single(), src/sigfish.c:330-350: pA = ((float)raw + offset) * (range / digitisation), all in fp32
void raw_to_picoamps(const int16_t *raw, int64_t n, double digitisation, double offset, double range, float *out);

// getevents(), src/events.c:557-577 -> detect_events() 510-554 (scrappie's t-statistic peak picker)
std::vector<sfa_event_t> detect_events(const float *pa, int64_t n, bool rna);

// RNA "-p -1": detect_query_start(), src/sigfish.c:380-422 (adaptor + poly-A segmenters, src/jnn.c)
// returns the first event index after the poly-A tail or -1
int64_t detect_query_start(const int16_t *raw, int64_t n, const float *pa, const std::vector<sfa_event_t> &ev, int pore);

// normalise_single(), src/sigfish.c:424-505: choose [qstart,qend), z-normalise the event means in place.
// Returns false when the read is dropped (et.n = 0 in the reference).  *status: 0 ok, 1 too short (kept),
// 2 ignored, |4 when the automatic prefix detection failed (fallback 50).
bool select_and_normalise(std::vector<sfa_event_t> &ev, const int16_t *raw, int64_t nraw, const float *pa, int32_t prefix_size,
                          int32_t query_size, uint32_t flag, int pore, int64_t *qstart, int64_t *qend, int *status);

}  // namespace sfa
