/* sdtw_oracle.c -- TEST INFRASTRUCTURE ONLY (see sdtw_oracle.h).
 *
 * A from-scratch CPU restatement of the algorithm behind `sigfish dtw`'s alignment stage.  Every function
 * cites the reference lines (relative to /root/reference) whose behaviour it restates.  It deliberately keeps
 * the reference's cost structure (a full qlen x rlen fp32 matrix per read/contig, traceback on every top-5
 * insertion, one pthread fan-out per batch) because it doubles as the "port" CPU baseline in bench.py.
 *
 * Parity pin: tests/test_oracle_vs_reference.py (needs oracle/_ref, i.e. this container) and
 * tests/test_oracle_golden.py (committed fixtures, runs anywhere).
 */
#define _GNU_SOURCE
#include "sdtw_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TOPN 5 /* SECONDARY_CAP, src/sigfish.h:41 */

/* ---- src/cdtw.c:25-36 : NaN-losing three-way minimum, evaluated a, then b, then c ---- */
static inline float low3(float a, float b, float c) {
    float m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return m;
}

/* ---- src/cdtw.c:171-189 ---- */
void orc_subsequence(const float *x, const float *y, int n, int m, float *cost) {
    float *row = cost;
    row[0] = fabs(x[0] - y[0]);
    for (int j = 1; j < m; j++) row[j] = fabs(x[0] - y[j]); /* free start along row 0 */
    for (int i = 1; i < n; i++) {
        const float *up = row;
        row += m;
        row[0] = fabs(x[i] - y[0]) + up[0]; /* column 0 is cumulative */
        for (int j = 1; j < m; j++) row[j] = fabs(x[i] - y[j]) + low3(up[j], up[j - 1], row[j - 1]);
    }
}

/* ---- src/cdtw.c:69-94 with squared==0 (the only use, src/sigfish.c:915) ---- */
float orc_std_dtw(const float *x, const float *y, int n, int m, float *cost) {
    float *row = cost;
    row[0] = fabs(x[0] - y[0]);
    for (int j = 1; j < m; j++) row[j] = fabs(x[0] - y[j]) + row[j - 1];
    for (int i = 1; i < n; i++) {
        const float *up = row;
        row += m;
        row[0] = fabs(x[i] - y[0]) + up[0];
        for (int j = 1; j < m; j++) row[j] = fabs(x[i] - y[j]) + low3(up[j], up[j - 1], row[j - 1]);
    }
    return cost[(size_t)n * m - 1];
}

/* ---- src/cdtw.c:98-167 (path) followed by 192-227 (drop the leading row-0 run) ----
 * Walk back from (n-1,starty): on row 0 go left, on column 0 go up, otherwise prefer the diagonal, then
 * left, then up, each tested by equality with the three-way minimum.  Returned in forward order. */
int orc_subsequence_path(const float *cost, int n, int m, int starty, int *px, int *py) {
    if (starty >= m) return 0;
    if (starty < 0) starty = m - 1;
    int i = n - 1, j = starty;
    int cap = n + starty + 1;
    int *bx = (int *)malloc(sizeof(int) * cap), *by = (int *)malloc(sizeof(int) * cap);
    int k = 0;
    bx[k] = i;
    by[k] = j;
    k++;
    while (i > 0 || j > 0) {
        if (i == 0) {
            j--;
        } else if (j == 0) {
            i--;
        } else {
            const float up = cost[(size_t)(i - 1) * m + j];
            const float dg = cost[(size_t)(i - 1) * m + (j - 1)];
            const float lf = cost[(size_t)i * m + (j - 1)];
            const float best = low3(up, dg, lf);
            if (dg == best) {
                i--;
                j--;
            } else if (lf == best) {
                j--;
            } else {
                i--;
            }
        }
        bx[k] = i;
        by[k] = j;
        k++;
    }
    /* forward order is bx[k-1] .. bx[0]; entries 1.. of the forward path that are still on row 0 are cut */
    int lead = 0;
    for (int f = 1; f < k; f++) {
        if (bx[k - 1 - f] == 0)
            lead++;
        else
            break;
    }
    int outk = k - lead;
    for (int f = 0; f < outk; f++) {
        px[f] = bx[k - 1 - lead - f];
        py[f] = by[k - 1 - lead - f];
    }
    free(bx);
    free(by);
    return outk;
}

int orc_path_start(const float *cost, int n, int m, int starty) {
    int eff = (starty < 0) ? m - 1 : starty;
    int *px = (int *)malloc(sizeof(int) * (n + eff + 2));
    int *py = (int *)malloc(sizeof(int) * (n + eff + 2));
    int k = orc_subsequence_path(cost, n, m, starty, px, py);
    int s = (k > 0) ? py[0] : -1;
    free(px);
    free(py);
    return s;
}

/* ---- src/genref.c:23-47 and src/sigfish.c:483-502: same arithmetic, sequential fp32 sums ---- */
void orc_normalise(float *v, uint64_t n) {
    float mean = 0, var = 0, sd = 0;
    float cnt = n;
    for (uint64_t j = 0; j < n; j++) mean += v[j];
    mean /= cnt;
    for (uint64_t j = 0; j < n; j++) var += (v[j] - mean) * (v[j] - mean);
    var /= cnt;
    sd = sqrt(var);
    for (uint64_t j = 0; j < n; j++) v[j] = (v[j] - mean) / sd;
}

/* ---- src/ref.h:13-41 (non-ACGT -> rank 0; the reference also prints a warning) ---- */
static inline uint32_t base_rank(char b) {
    switch (b) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}
uint32_t orc_kmer_rank(const char *s, uint32_t k) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < k; i++) r += base_rank(s[k - 1 - i]) << (2 * i);
    return r;
}
/* src/ref.h:43-68: A<->T, C<->G (either case); anything else complements to 'T' */
static inline char comp(char c) {
    switch (c) {
        case 'A': case 'a': return 'T';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        case 'T': case 't': return 'A';
        default: return 'T';
    }
}

/* ---- src/genref.c:127-217 for one record ---- */
int32_t orc_gen_ref_record(const char *seq, int32_t l, const float *level_mean, uint32_t k, uint32_t flag,
                           int32_t query_size, float *fwd, float *rev, int32_t *st_offset) {
    const int rna = (flag & ORC_RNA) != 0;
    int32_t ref_len;
    if (!rna || (flag & ORC_REF)) {
        ref_len = l + 1 - k;
    } else {
        uint32_t heu = query_size * 1.5; /* genref.c:133 */
        uint32_t full = (uint32_t)l + 1 - k;
        ref_len = heu > full ? full : heu;
    }
    *st_offset = 0;
    if (!rna) {
        char *rc = (char *)malloc((size_t)l + 1);
        for (int32_t i = 0; i < l; i++) rc[i] = comp(seq[l - 1 - i]);
        rc[l] = 0;
        for (int32_t j = 0; j < ref_len; j++) {
            fwd[j] = level_mean[orc_kmer_rank(seq + j, k)];
            rev[j] = level_mean[orc_kmer_rank(rc + j, k)];
        }
        free(rc);
    } else if (flag & ORC_INV) { /* genref.c:166-177 */
        const char *tail = seq + l - ref_len - (k - 1);
        for (int32_t j = 0; j < ref_len; j++) fwd[ref_len - j - 1] = level_mean[orc_kmer_rank(tail + j, k)];
    } else {
        const char *st;
        if (flag & ORC_END) {
            st = seq;
        } else {
            st = seq + l - ref_len - (k - 1);
            *st_offset = l - ref_len - (k - 1);
        }
        for (int32_t j = 0; j < ref_len; j++) fwd[j] = level_mean[orc_kmer_rank(st + j, k)];
    }
    orc_normalise(fwd, ref_len);
    if (!rna) orc_normalise(rev, ref_len);
    return ref_len;
}

/* ---- src/sigfish.c:507-626: sorted top-5 list, worst first; a traceback on every insertion ---- */
typedef struct {
    int32_t rid, pos_st, pos_end;
    float score;
    char d;
} cand_t;

static void top_init(cand_t *t) {
    for (int l = 0; l < TOPN; l++) {
        t[l].rid = -1;
        t[l].pos_st = -1;
        t[l].pos_end = -1;
        t[l].score = INFINITY;
        t[l].d = 0;
    }
}

static void top_offer(cand_t *t, float score, int32_t rid, int32_t pos, char d, const float *cost, int32_t qlen,
                      int32_t rlen) {
    int l = 0;
    while (l < TOPN && !(score > t[l].score)) l++; /* sigfish.c:577-583: ties keep scanning */
    if (l == 0) return;
    for (int m = 0; m < l - 1; m++) t[m] = t[m + 1];
    cand_t *slot = &t[l - 1];
    slot->score = score;
    slot->pos_end = pos;
    slot->rid = rid;
    slot->d = d;
    slot->pos_st = orc_path_start(cost, qlen, rlen, pos); /* sigfish.c:599-604 */
}

/* sigfish.c:891-901 / 938-948: one candidate per window of qlen columns on the last row */
static void scan_windows(cand_t *t, const float *cost, int32_t qlen, int32_t rlen, int32_t rid, char d) {
    const float *last = cost + (size_t)(qlen - 1) * rlen;
    for (int32_t w = 0; w < rlen; w += qlen) {
        float best = INFINITY;
        int32_t at = -1;
        for (int32_t m = 0; m < qlen && w + m < rlen; m++) {
            if (last[w + m] < best) {
                best = last[w + m];
                at = w + m;
            }
        }
        /* the reference passes min_pos-(qlen-1)*rlen with min_pos=-1 when nothing was finite */
        int32_t pos = (at >= 0) ? at : (int32_t)(-1 - (int64_t)(qlen - 1) * rlen);
        top_offer(t, best, rid, pos, d, cost, qlen, rlen);
    }
}

/* ---- src/sigfish.c:979-983: (int)round(...) saturating like x86 cvttsd2si, capped at 60, stored in u8 ---- */
uint8_t orc_mapq(float score, float score2) {
    float v = 500 * (score2 - score) / score;
    double r = round(v);
    int q;
    if (!(r >= -2147483648.0 && r <= 2147483647.0))
        q = INT_MIN;
    else
        q = (int)r;
    if (q > 60) q = 60;
    return (uint8_t)q;
}

/* ---- src/sigfish.c:828-992 ---- */
void orc_dtw_single(const float *events, int32_t qlen, const orc_ref_t *ref, uint32_t flag, orc_result_t *out) {
    memset(out, 0, sizeof(*out));
    if (qlen <= 0) return; /* caller encodes "et.n==0" as qlen 0 */
    const int rna = (flag & ORC_RNA) != 0;
    cand_t top[TOPN];
    top_init(top);

    float *query = (float *)malloc(sizeof(float) * qlen);
    for (int32_t j = 0; j < qlen; j++) {
        if (rna && !(flag & ORC_INV))
            query[qlen - 1 - j] = events[j]; /* sigfish.c:861-863 */
        else
            query[j] = events[j];
    }
    for (int32_t c = 0; c < ref->num_ref; c++) {
        const int32_t rlen = ref->ref_lengths[c];
        float *cost = (float *)malloc(sizeof(float) * (size_t)qlen * rlen);
        if (!(flag & ORC_DTW)) {
            orc_subsequence(query, ref->forward[c], qlen, rlen, cost);
            scan_windows(top, cost, qlen, rlen, c, '+');
        } else {
            orc_std_dtw(query, ref->forward[c], qlen, rlen, cost);
            top_offer(top, cost[(size_t)qlen * rlen - 1], c, rlen - 1, '+', cost, qlen, rlen);
        }
        if (!rna) {
            orc_subsequence(query, ref->reverse[c], qlen, rlen, cost);
            scan_windows(top, cost, qlen, rlen, c, '-');
        }
        free(cost);
    }
    free(query);

    const cand_t *b = &top[TOPN - 1];
    out->score = b->score;
    out->score2 = top[TOPN - 2].score;
    if (b->rid < 0) { /* nothing was inserted: the reference would index ref arrays with -1 (UB) */
        out->rid = -1;
        out->pos_st = out->pos_end = -1;
        out->strand = 0;
        out->mapq = 0;
        out->valid = 1;
        return;
    }
    const int32_t rl = ref->ref_lengths[b->rid];
    out->pos_st = (b->d == '+') ? b->pos_st : rl - b->pos_end; /* sigfish.c:971-972 */
    out->pos_end = (b->d == '+') ? b->pos_end : rl - b->pos_st;
    out->pos_st += ref->ref_st_offset[b->rid];
    out->pos_end += ref->ref_st_offset[b->rid];
    out->rid = b->rid;
    out->strand = b->d;
    out->mapq = orc_mapq(out->score, out->score2);
    out->valid = 1;
}

/* ---- src/thread.c:24-132: static split into t ranges + atomic pops + steal from the first busy range ---- */
typedef struct worker_s {
    const float *events;
    const int64_t *q_off;
    const orc_ref_t *ref;
    uint32_t flag;
    orc_result_t *out;
    int32_t next, end;
    struct worker_s *all;
    int32_t nworkers;
} worker_t;

static void run_read(worker_t *w, int32_t i) {
    orc_dtw_single(w->events + w->q_off[i], (int32_t)(w->q_off[i + 1] - w->q_off[i]), w->ref, w->flag, &w->out[i]);
}

static void *worker_main(void *arg) {
    worker_t *w = (worker_t *)arg;
    for (;;) {
        int32_t i = __sync_fetch_and_add(&w->next, 1);
        if (i >= w->end) break;
        run_read(w, i);
    }
    for (;;) { /* steal: first worker with more than one item left (STEAL_THRESH 1) */
        worker_t *victim = NULL;
        for (int32_t v = 0; v < w->nworkers; v++) {
            if (w->all[v].end - w->all[v].next > 1) {
                victim = &w->all[v];
                break;
            }
        }
        if (!victim) break;
        int32_t i = __sync_fetch_and_add(&victim->next, 1);
        if (i >= victim->end) break;
        run_read(w, i);
    }
    return NULL;
}

void orc_align_batch(const float *events, const int64_t *q_off, int32_t n_reads, const orc_ref_t *ref,
                     uint32_t flag, int32_t num_thread, orc_result_t *out) {
    if (num_thread <= 1) {
        for (int32_t i = 0; i < n_reads; i++)
            orc_dtw_single(events + q_off[i], (int32_t)(q_off[i + 1] - q_off[i]), ref, flag, &out[i]);
        return;
    }
    worker_t *ws = (worker_t *)calloc(num_thread, sizeof(worker_t));
    pthread_t *tid = (pthread_t *)calloc(num_thread, sizeof(pthread_t));
    int32_t step = (n_reads + num_thread - 1) / num_thread, at = 0;
    for (int32_t t = 0; t < num_thread; t++) {
        ws[t].events = events;
        ws[t].q_off = q_off;
        ws[t].ref = ref;
        ws[t].flag = flag;
        ws[t].out = out;
        ws[t].next = at < n_reads ? at : n_reads;
        at += step;
        ws[t].end = at > n_reads ? n_reads : at;
        ws[t].all = ws;
        ws[t].nworkers = num_thread;
    }
    for (int32_t t = 0; t < num_thread; t++) pthread_create(&tid[t], NULL, worker_main, &ws[t]);
    for (int32_t t = 0; t < num_thread; t++) pthread_join(tid[t], NULL);
    free(ws);
    free(tid);
}

/* ---- src/sigfish.c:628-660 ---- */
int orc_paf_row(char *buf, int cap, const orc_result_t *r, const char *read_id, const char *rname,
                uint64_t start_raw_idx, uint64_t end_raw_idx, uint64_t query_size, uint64_t len_raw_signal,
                uint64_t rlength) {
    float block_len = r->pos_end - r->pos_st;
    float residue = block_len - r->score * block_len / (query_size);
    return snprintf(buf, cap, "%s\t%ld\t%ld\t%ld\t%c\t%s\t%d\t%d\t%d\t%d\t%d\t%d\ttp:A:P\td1:f:%.2f\td2:f:%.2f\n", read_id,
                    (long)len_raw_signal, (long)start_raw_idx, (long)end_raw_idx, r->strand, rname, (int)rlength,
                    r->pos_st, r->pos_end, (int)round(residue), (int)round(block_len), r->mapq, r->score, r->score2);
}

/* ---- src/sigfish.c:433-480: which events form the query ---- */
int orc_query_window(int64_t n, int32_t prefix_size, int32_t query_size, uint32_t flag, int64_t *qstart,
                     int64_t *qend) {
    int64_t st, en;
    int keep = 1;
    if (!(flag & ORC_END)) {
        st = prefix_size; /* prefix_size<0 (auto) is resolved by the caller (jnn), fallback 50 */
        en = st + query_size;
        if (st + 25 > n) {
            st = en = 0;
            keep = 0;
        } else if (en > n) {
            en = n;
        }
    } else {
        st = n - prefix_size - query_size;
        en = n - prefix_size;
        if (st < 0) st = 0;
        if (en < 0) {
            en = 0;
            keep = 0;
        }
    }
    *qstart = st;
    *qend = en;
    return keep;
}
