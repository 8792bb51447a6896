/* ref_driver.c -- TEST INFRASTRUCTURE ONLY.  Drives the *compiled reference* (oracle/_ref/libsigfish_ref.so,
 * built by oracle/Makefile from /root/reference/src where the sources lie) through its own batch pipeline:
 * load_db -> process_db (parse, events, normalise, dtw_single) -> output_db, exactly like dtw_main.c:299-326,
 * and dumps the hot-path intermediates for golden fixtures.
 *
 * The one reference unit that cannot be built here is src/model.c (it includes the missing blob src/model.h),
 * so this driver does what init_core (sigfish.c:81-207) does minus the model.c calls: the k-mer level table is
 * DATA read from a float32 file (the --kmer-model path of the reference supplies the same table from text).
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sigfish.h" /* from /root/reference/src via -I */
#include "error.h"

/* non-static reference functions without a public prototype */
int8_t drna_detect(slow5_file_t *sp);
refsynth_t *gen_ref(const char *genome, model_t *pore_model, uint32_t kmer_size, uint32_t flag, int32_t query_size);
void free_ref(refsynth_t *ref);
void free_db(db_t *db);
int eval_main(int argc, char *argv[]); /* src/eval.c:380 */
#ifdef HAVE_ACC /* `make ref-acc`: the reference built with acc=1 + oracle/ref_acc.patch; the teardown slot of free_core() */
void sigfish_acc_free(void);
#endif

static void wr(FILE *f, const void *p, size_t n) { fwrite(p, 1, n, f); }

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "eval") == 0) return eval_main(argc - 1, argv + 1); /* the reference's `sigfish eval` */
    static struct option lo[] = {{"rna", 0, 0, 1},      {"dtw-std", 0, 0, 2}, {"invert", 0, 0, 3},
                                 {"full-ref", 0, 0, 4}, {"from-end", 0, 0, 5}, {"sam", 0, 0, 6},
                                 {"dump", 1, 0, 7},     {"model", 1, 0, 8},    {"kmer", 1, 0, 9},
                                 {"profile-cpu", 0, 0, 10}, {0, 0, 0, 0}};
    opt_t opt;
    init_opt(&opt);
    const char *dump = NULL, *model = NULL;
    int k = 0, c, li;
    while ((c = getopt_long(argc, argv, "q:p:t:K:", lo, &li)) >= 0) {
        switch (c) {
            case 'q': opt.query_size = atoi(optarg); break;
            case 'p': opt.prefix_size = atoi(optarg); break;
            case 't': opt.num_thread = atoi(optarg); break;
            case 'K': opt.batch_size = atoi(optarg); break;
            case 1: opt.flag |= SIGFISH_RNA; break;
            case 2: opt.flag |= SIGFISH_DTW; break;
            case 3: opt.flag |= SIGFISH_INV; break;
            case 4: opt.flag |= SIGFISH_REF; break;
            case 5: opt.flag |= SIGFISH_END; break;
            case 6: opt.flag |= SIGFISH_SAM; break;
            case 7: dump = optarg; break;
            case 8: model = optarg; break;
            case 9: k = atoi(optarg); break;
            case 10: opt.flag |= SIGFISH_PRF; break;
            default: return 2;
        }
    }
    if (argc - optind != 2 || !model || k <= 0) {
        fprintf(stderr, "usage: ref_driver --model levels.f32 --kmer K [opts] ref.fa reads.blow5\n");
        return 2;
    }
    set_log_level(LOG_ERR);
    core_t *core = (core_t *)calloc(1, sizeof(core_t));
    core->sf = slow5_open(argv[optind + 1], "r");
    if (!core->sf) return 3;
    if (drna_detect(core->sf)) opt.flag |= SIGFISH_RNA;
    core->model = (model_t *)calloc(MAX_NUM_KMER, sizeof(model_t));
    {
        FILE *mf = fopen(model, "rb");
        if (!mf) return 4;
        size_t nk = (size_t)1 << (2 * k);
        float *lv = (float *)malloc(nk * sizeof(float));
        if (fread(lv, sizeof(float), nk, mf) != nk) return 5;
        fclose(mf);
        for (size_t i = 0; i < nk; i++) {
            core->model[i].level_mean = lv[i];
            core->model[i].level_stdv = 1.5f;
        }
        free(lv);
    }
    core->kmer_size = k;
    core->ref = gen_ref(argv[optind], core->model, k, opt.flag, opt.query_size);
    core->opt = opt;

    FILE *df = NULL;
    if (dump) {
        df = fopen(dump, "wb");
        int8_t rna = (opt.flag & SIGFISH_RNA) ? 1 : 0;
        int32_t hdr[4] = {0x53464131, core->ref->num_ref, rna, (int32_t)opt.flag};
        wr(df, hdr, sizeof hdr);
        for (int i = 0; i < core->ref->num_ref; i++) {
            int32_t nl = strlen(core->ref->ref_names[i]);
            int32_t m[4] = {core->ref->ref_lengths[i], core->ref->ref_seq_lengths[i], core->ref->ref_st_offset[i], nl};
            wr(df, m, sizeof m);
            wr(df, core->ref->ref_names[i], nl);
            wr(df, core->ref->forward[i], sizeof(float) * m[0]);
            if (!rna) wr(df, core->ref->reverse[i], sizeof(float) * m[0]);
        }
    }
    if (opt.flag & SIGFISH_SAM) {
        for (int i = 0; i < core->ref->num_ref; i++)
            fprintf(stdout, "@SQ\tSN:%s\tLN:%ld\n", core->ref->ref_names[i], (long)core->ref->ref_lengths[i]);
        fprintf(stdout, "@PG\tID:sigfish\tPN:sigfish\tVN:%s\n", SIGFISH_VERSION);
    }
    db_t *db = init_db(core);
    ret_status_t st = {core->opt.batch_size, core->opt.batch_size_bytes};
    while (st.num_reads >= core->opt.batch_size || st.num_bytes >= core->opt.batch_size_bytes) {
        st = load_db(core, db);
        process_db(core, db);
        output_db(core, db);
        if (df) {
            for (int i = 0; i < db->n_rec; i++) {
                slow5_rec_t *r = db->slow5_rec[i];
                int32_t idl = strlen(r->read_id);
                int8_t valid = (r->len_raw_signal > 0 && db->et[i].n > 0);
                wr(df, &idl, 4);
                wr(df, r->read_id, idl);
                int64_t v[4] = {(int64_t)r->len_raw_signal, (int64_t)db->et[i].n, valid ? db->qstart[i] : 0,
                                valid ? db->qend[i] : 0};
                wr(df, v, sizeof v);
                wr(df, &valid, 1);
                if (!valid) continue;
                int64_t qs = db->qstart[i], qe = db->qend[i];
                uint64_t s0 = db->et[i].event[qs].start, s1 = db->et[i].event[qe - 1].start;
                float l1 = db->et[i].event[qe - 1].length;
                wr(df, &s0, 8);
                wr(df, &s1, 8);
                wr(df, &l1, 4);
                for (int64_t j = qs; j < qe; j++) wr(df, &db->et[i].event[j].mean, 4);
                aln_t *a = &db->aln[i];
                wr(df, &a->rid, 4);
                wr(df, &a->pos_st, 4);
                wr(df, &a->pos_end, 4);
                wr(df, &a->score, 4);
                wr(df, &a->score2, 4);
                wr(df, &a->d, 1);
                wr(df, &a->mapq, 1);
            }
        }
        free_db_tmp(db);
    }
    if (df) fclose(df);
    if (opt.flag & SIGFISH_PRF)
        fprintf(stderr, "parse %.4f events %.4f normalise %.4f dtw %.4f\n", core->parse_time, core->event_time,
                core->normalise_time, core->dtw_time);
    free_db(db);
#ifdef HAVE_ACC
    if (core->opt.flag & SIGFISH_ACC) sigfish_acc_free(); /* free_core(), src/sigfish.c:221-225 (not callable here: model.c is absent) */
#endif
    free_ref(core->ref);
    slow5_close(core->sf);
    free(core->model);
    free(core);
    return 0;
}
