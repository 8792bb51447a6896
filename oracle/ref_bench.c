/* ref_bench.c -- TEST INFRASTRUCTURE ONLY.  Times the COMPILED REFERENCE's own alignment stage (dtw_single fanned out
 * by work_db, i.e. align_db, src/sigfish.c:1003-1015 + src/thread.c) on queries given at the stage boundary, so that
 * bench.py can report sigfish's CPU path next to the GPU number on the same box ("cpu_baseline.kind": "reference").
 * Built by oracle/Makefile against oracle/_ref/libsigfish_ref.so; reference headers come from /root/reference via -I.
 *
 * input file : int32 flag, num_ref, n_reads, num_thread; per contig: int32 ref_len, seq_len, st_offset; float fwd[ref_len],
 *              [float rev[ref_len] unless RNA]; int64 q_off[n_reads+1]; float queries[q_off[n]]  (event order, normalised)
 * output file: double seconds; per read: int32 rid,pos_st,pos_end; float score,score2; int8 d; uint8 mapq; uint8 valid; uint8 0
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "sigfish.h"
#include "error.h"

void dtw_single(core_t *core, db_t *db, int32_t i);

static double now(void) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}
#define RD(p, n) if (fread((p), 1, (n), f) != (size_t)(n)) { fprintf(stderr, "short read\n"); return 2; }

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: ref_bench in.bin out.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    RD(hdr, sizeof hdr);
    const int32_t flag = hdr[0], num_ref = hdr[1], n = hdr[2], nth = hdr[3];
    const int rna = flag & SIGFISH_RNA;
    set_log_level(LOG_ERR);
    core_t *core = (core_t *)calloc(1, sizeof(core_t));
    init_opt(&core->opt);
    core->opt.flag = flag;
    core->opt.num_thread = nth;
    core->opt.batch_size = n;
    refsynth_t *ref = (refsynth_t *)calloc(1, sizeof(refsynth_t));
    ref->num_ref = num_ref;
    ref->ref_names = (char **)calloc(num_ref, sizeof(char *));
    ref->ref_lengths = (int32_t *)calloc(num_ref, 4);
    ref->ref_seq_lengths = (int32_t *)calloc(num_ref, 4);
    ref->ref_st_offset = (int32_t *)calloc(num_ref, 4);
    ref->forward = (float **)calloc(num_ref, sizeof(float *));
    ref->reverse = rna ? NULL : (float **)calloc(num_ref, sizeof(float *));
    for (int i = 0; i < num_ref; i++) {
        int32_t m[3];
        RD(m, sizeof m);
        ref->ref_lengths[i] = m[0];
        ref->ref_seq_lengths[i] = m[1];
        ref->ref_st_offset[i] = m[2];
        ref->ref_names[i] = (char *)malloc(32);
        snprintf(ref->ref_names[i], 32, "contig%d", i);
        ref->forward[i] = (float *)malloc(sizeof(float) * m[0]);
        RD(ref->forward[i], sizeof(float) * m[0]);
        if (!rna) {
            ref->reverse[i] = (float *)malloc(sizeof(float) * m[0]);
            RD(ref->reverse[i], sizeof(float) * m[0]);
        }
    }
    core->ref = ref;
    int64_t *q_off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    RD(q_off, sizeof(int64_t) * (n + 1));
    float *q = (float *)malloc(sizeof(float) * (q_off[n] > 0 ? q_off[n] : 1));
    RD(q, sizeof(float) * q_off[n]);
    fclose(f);

    db_t *db = init_db(core);
    db->n_rec = n;
    for (int i = 0; i < n; i++) {
        const int64_t len = q_off[i + 1] - q_off[i];
        slow5_rec_t *r = (slow5_rec_t *)calloc(1, sizeof(slow5_rec_t));
        r->read_id = (char *)malloc(24);
        snprintf(r->read_id, 24, "read%d", i);
        r->len_raw_signal = len > 0 ? 10 * (uint64_t)len + 10 : 0;
        db->slow5_rec[i] = r;
        db->et[i].n = len;
        db->et[i].event = (event_t *)calloc(len > 0 ? len : 1, sizeof(event_t));
        for (int64_t j = 0; j < len; j++) {
            db->et[i].event[j].mean = q[q_off[i] + j];
            db->et[i].event[j].start = 10 * j;
            db->et[i].event[j].length = 10;
        }
        db->qstart[i] = 0;
        db->qend[i] = len;
        memset(&db->aln[i], 0, sizeof(aln_t));
    }
    const double t0 = now();
    work_db(core, db, dtw_single); /* = align_db() */
    const double dt = now() - t0;
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(&dt, 8, 1, o);
    for (int i = 0; i < n; i++) {
        aln_t *a = &db->aln[i];
        const uint8_t valid = db->et[i].n > 0;
        const uint8_t zero = 0;
        fwrite(&a->rid, 4, 1, o);
        fwrite(&a->pos_st, 4, 1, o);
        fwrite(&a->pos_end, 4, 1, o);
        fwrite(&a->score, 4, 1, o);
        fwrite(&a->score2, 4, 1, o);
        fwrite(&a->d, 1, 1, o);
        fwrite(&a->mapq, 1, 1, o);
        fwrite(&valid, 1, 1, o);
        fwrite(&zero, 1, 1, o);
    }
    fclose(o);
    fprintf(stderr, "reference align_db: %d reads, %d threads, %.3f s\n", n, nth, dt);
    return 0;
}
