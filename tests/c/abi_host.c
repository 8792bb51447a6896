/* abi_host.c -- a plain C99 host calling the C-ABI the way the reference's align_db hook would:
 * sfa_init with refsynth_t-shaped arrays, sfa_align_events with db_t-shaped event tables, sfa_destroy.
 * usage: abi_host in.bin out.bin     (files written / read by tests/test_c_host.py)
 * in : int32 flag, num_ref, n_reads; per contig int32 ref_len, st_offset, float fwd[ref_len][, float rev[ref_len]];
 *      per read int64 n_events, qstart, qend, float mean[n_events]
 * out: sfa_result_t[n_reads] */
#include <stdio.h>
#include <stdlib.h>

#include "sigfish_amd.h"

#define RD(p, n) if (fread((p), 1, (n), f) != (size_t)(n)) { fprintf(stderr, "short read\n"); return 2; }

int main(int argc, char **argv) {
    FILE *f;
    int32_t hdr[3], i;
    if (argc != 3 || !(f = fopen(argv[1], "rb"))) return 2;
    RD(hdr, sizeof hdr);
    {
        const int32_t flag = hdr[0], num_ref = hdr[1], n = hdr[2];
        const int rna = flag & SFA_RNA;
        int32_t *ref_len = malloc(sizeof(int32_t) * num_ref), *ref_off = malloc(sizeof(int32_t) * num_ref);
        float **fwd = malloc(sizeof(float *) * num_ref), **rev = malloc(sizeof(float *) * num_ref);
        sfa_event_t **ev = malloc(sizeof(sfa_event_t *) * n);
        int64_t *nev = malloc(sizeof(int64_t) * n), *qs = malloc(sizeof(int64_t) * n), *qe = malloc(sizeof(int64_t) * n);
        sfa_result_t *rows = malloc(sizeof(sfa_result_t) * n);
        sfa_ref_t ref;
        sfa_ctx_t *ctx = NULL;
        FILE *o;
        for (i = 0; i < num_ref; i++) {
            int32_t m[2];
            RD(m, sizeof m);
            ref_len[i] = m[0];
            ref_off[i] = m[1];
            fwd[i] = malloc(sizeof(float) * m[0]);
            RD(fwd[i], sizeof(float) * m[0]);
            rev[i] = NULL;
            if (!rna) {
                rev[i] = malloc(sizeof(float) * m[0]);
                RD(rev[i], sizeof(float) * m[0]);
            }
        }
        for (i = 0; i < n; i++) {
            int64_t m[3], j;
            RD(m, sizeof m);
            nev[i] = m[0];
            qs[i] = m[1];
            qe[i] = m[2];
            ev[i] = calloc(m[0] > 0 ? m[0] : 1, sizeof(sfa_event_t));
            for (j = 0; j < m[0]; j++) {
                RD(&ev[i][j].mean, sizeof(float));
                ev[i][j].start = 7 * j;
                ev[i][j].length = 7;
            }
        }
        fclose(f);
        ref.num_ref = num_ref;
        ref.ref_lengths = ref_len;
        ref.ref_st_offset = ref_off;
        ref.forward = (const float *const *)fwd;
        ref.reverse = rna ? NULL : (const float *const *)rev;
        if (sfa_init(&ctx, &ref, (uint32_t)flag, 0) != SFA_OK) {
            fprintf(stderr, "sfa_init: %s\n", sfa_last_error());
            return 1;
        }
        if (sfa_align_events(ctx, (const sfa_event_t *const *)ev, nev, qs, qe, n, rows) != SFA_OK) {
            fprintf(stderr, "sfa_align_events: %s\n", sfa_last_error());
            return 1;
        }
        sfa_destroy(ctx);
        if (!(o = fopen(argv[2], "wb"))) return 2;
        fwrite(rows, sizeof(sfa_result_t), n, o);
        fclose(o);
    }
    return 0;
}
