// strip_rows_host.hip -- host-side check of sfa::strip_rows_per_lane (sdtw_strips.hpp): for EVERY query length the row-strip
// kernels may see, the strips of a query (ceil(qlen / 2048) of them, 64 lanes x R rows each) cover the query, the last strip
// holds at least one row, and R is one of the instantiated heights.  Built with hipcc (the header is HIP), runs on the CPU.
#include <cstdio>
#include <cstdlib>

#include "sdtw_strips.hpp"

int main(int argc, char **argv) {
    const int max_q = argc > 1 ? atoi(argv[1]) : 400000;
    long bad = 0;
    double worst = 1.0;
    for (int q = sfa::kStripRows + 1; q <= max_q; ++q) {  // (queries of up to 2048 events never come here)
        const int n = (q + sfa::kStripRows - 1) / sfa::kStripRows;
        const int R = sfa::strip_rows_per_lane(q);
        const bool ok = (R == 20 || R == 24 || R == 28 || R == 32) && 64L * R * n >= q && 64L * R * (n - 1) < q;
        if (!ok && bad++ < 10) fprintf(stderr, "qlen %d: R %d, %d strips\n", q, R, n);
        const double fill = static_cast<double>(q) / (64.0 * R * n);
        if (fill < worst) worst = fill;
    }
    printf("%ld violations; least filled strips: %.3f of their rows are query rows\n", bad, worst);
    return bad != 0;
}
