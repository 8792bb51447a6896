/* sigfish_amd.h -- C-ABI of the MI355X-native sDTW alignment stage.
 *
 * Drop-in boundary for the reference's accelerator hook (all citations relative to hasindu2008/sigfish v0.2.0):
 *
 *   reference site                                   this library
 *   -----------------------------------------------  -----------------------------------------------
 *   init slot      src/sigfish.c:200-204 (HAVE_ACC)  sfa_init()          upload reference event arrays
 *   align_db()     src/sigfish.c:1003-1015           sfa_align_events()  whole batch, after normalise stage
 *                                                    sfa_align_batch()   same, packed SoA queries
 *   teardown slot  src/sigfish.c:221-225             sfa_destroy()
 *
 * Plain C types only (no torch / HIP types); every pointer is caller-owned unless stated.  All functions return
 * 0 on success, a negative SFA_E* code otherwise; sfa_last_error() gives the message.  Nothing here falls back to
 * a CPU implementation: without a usable gfx950 device sfa_init() fails.
 */
#ifndef SIGFISH_AMD_H
#define SIGFISH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFA_VERSION "0.1.0"

/* option bits: numerically identical to the reference's opt.flag (src/sigfish.h:30-39) so that a host can pass
 * core->opt.flag straight through.  Bits not listed are ignored. */
#define SFA_RNA 0x001 /* SIGFISH_RNA: single strand, query reversed (src/sigfish.c:860-866) */
#define SFA_DTW 0x002 /* SIGFISH_DTW: --dtw-std, standard DTW instead of subsequence (src/sigfish.c:914-917) */
#define SFA_INV 0x004 /* SIGFISH_INV: --invert, query NOT reversed for RNA */
#define SFA_REF 0x010 /* SIGFISH_REF: --full-ref (only affects how the caller built the reference arrays) */
#define SFA_END 0x020 /* SIGFISH_END: --from-end (only affects which events the caller hands over) */

enum {
    SFA_OK = 0,
    SFA_EINVAL = -1,  /* bad argument */
    SFA_ENODEV = -2,  /* no usable GPU / HIP error */
    SFA_ENOMEM = -3,  /* allocation failed */
    SFA_ERANGE = -4,  /* a size out of range (output buffer too small; sfa_plan_batch: query beyond SFA_MAX_QUERY) */
    SFA_EKERNEL = -5  /* kernel launch or execution failed (incl. an in-launch hand-over that did not arrive within "spin_limit_ms") */
};

/* Queries of up to SFA_MAX_QUERY events are held in the registers of one wavefront (two-pass kernels).  Longer ones --
 * the reference has no limit on -q (src/cdtw.c:171-189 fills whatever it is given) -- are accepted by every align entry
 * point and run as row strips of SFA_MAX_QUERY query rows each (sdtw_strips.hpp), in the same call, same rows. */
#define SFA_MAX_QUERY 2048

/* Reference event model: the fields of refsynth_t (src/sigfish.h:90-99) the alignment stage reads. */
typedef struct {
    int32_t num_ref;
    const int32_t *ref_lengths;   /* [num_ref] k-mer counts (ref->ref_lengths) */
    const int32_t *ref_st_offset; /* [num_ref] (ref->ref_st_offset) */
    const float *const *forward;  /* [num_ref][ref_lengths[i]] z-normalised expected levels */
    const float *const *reverse;  /* same for the reverse complement; NULL when SFA_RNA */
} sfa_ref_t;

/* One result row per read: the fields of aln_t (src/sigfish.h:146-158) that output formatting consumes. */
typedef struct {
    int32_t rid;     /* contig index, -1 if nothing aligned */
    int32_t pos_st;  /* after strand flip and ref_st_offset (src/sigfish.c:971-975) */
    int32_t pos_end;
    float score;     /* best   (d1) */
    float score2;    /* second (d2), +inf when there was a single candidate */
    int8_t strand;   /* '+' or '-' */
    uint8_t mapq;    /* src/sigfish.c:979-983 */
    uint8_t valid;   /* 0: read skipped (no events) -- the reference prints nothing for it */
    uint8_t pad;
} sfa_result_t;

/* The event record the reference batches hold (event_t, src/sigfish.h:57-64); used by sfa_align_events. */
typedef struct {
    uint64_t start;
    float length;
    float mean;
    float stdv;
} sfa_event_t;

typedef struct sfa_ctx sfa_ctx_t;

/* Device-side timing of the last call (HIP events recorded on the stream the kernels run on). */
typedef struct {
    double fill_ms;        /* pass 1: sdtw_fill_kernel (the dominant kernel).  A batch with queries beyond 2048 events runs their
                              row strips on streams of their own beside the other kernels: the stages then overlap, fill_ms is the
                              device time of the whole batch (= total_ms) and trace_ms is 0 */
    double trace_ms;       /* pass 2: sdtw_trace_kernel (start-column recovery of the winners) */
    double finalize_ms;    /* per-read reductions / row assembly */
    double total_ms;       /* first kernel start -> last kernel end */
    int64_t cells;         /* DP cells of the batch, algorithmic: sum(qlen) * sum over (contig,strand) of rlen */
    int64_t fill_launches; /* fill launches in the call (1) */
    int64_t ckpt_interval; /* steps between the snapshots of pass 1 in HBM (with lds_ckpt: of the sparse store) */
    int64_t ckpt_bytes;    /* HBM taken by the checkpoints of the batch */
    int64_t n_tasks;       /* wave-tasks of the fill launch */
    int64_t n_chunks;      /* pieces the (contig,strand) list was cut into */
    int64_t n_segments;    /* column segments per (contig,strand) (1: off; small batches use several, verified) */
    int64_t segment_reruns; /* batches of this context walked again because a segment hand-over did not verify */
    /* sfa_align_raw only (0 otherwise): the stages in front of the alignment, the reference's "Events time" and
     * "Normalise time" (src/sigfish.c:1026-1035) as device time */
    double events_ms;      /* pA conversion, prefix sums, t-statistics, peak picking, event statistics */
    double normalise_ms;   /* query windows: z-normalisation + packing (the window choice itself is host arithmetic) */
    /* reads of the batch whose query holds a NaN or +-inf event.  The reference aborts on such a read (assert in update_aln,
     * src/sigfish.c:611); here they are skipped: their rows come back with valid = 0. */
    int64_t non_finite_reads;
    double decode_ms;      /* sfa_align_blow5 only: record decompression (inflate), field parsing and signal decoding on the device */
    int64_t blow5_fallbacks; /* batches of this context handed to the host reader because the device declined a record */
    int64_t lds_ckpt;      /* 1: the fill kept its rolling checkpoints in LDS (ckpt_interval is then the sparse HBM store's);
                              2: and pass 2 ran inside the fill launch; 0: every snapshot went to HBM */
    int64_t trace_margin;  /* head start of pass 2 in steps (a whole query + lanes, or less: see "trace_margin") */
    int64_t fused_trace;   /* 1: pass 2 ran inside the fill launch by ticket (fill_ms covers both, trace_ms is 0) -- the LDS-checkpoint
                              fill or the 32-row fill */
} sfa_profile_t;

/* How a batch is laid out on the device (host logic only; needs no GPU). */
typedef struct {
    int32_t n_quads;          /* wavefronts' worth of reads: groups of <= 4 reads with the same query length (<= 2 / 1
                                 reads for queries longer than 512 / 1024 events) */
    int32_t n_chunks;
    int32_t n_classes;        /* rows-per-lane classes present */
    int32_t max_rows_per_lane;
    int32_t ckpt_interval;
    int32_t trace_margin;
    int64_t ckpt_bytes;
    int64_t n_tasks;
    int32_t max_lanes_per_read; /* 16; 32 / 64 for queries longer than 512 / 1024 events or under lane widening */
    int32_t lane_widening;      /* 1, 2 or 4: small batches trade rows per lane for lanes per read */
} sfa_plan_info_t;

/* Create a context on HIP device `device`, copy the reference event arrays into HBM.
 * flag: SFA_* bits.  The arrays behind `ref` may be freed after the call returns. */
int sfa_init(sfa_ctx_t **ctx, const sfa_ref_t *ref, uint32_t flag, int device);

/* The same context over SEVERAL devices of one node (multi-GPU behind this ABI): reads shard
 * embarrassingly, so the context owns one ordinary context per entry of devices[] and every align call below cuts its
 * batch into contiguous read ranges [r*n/G, (r+1)*n/G), runs them side by side (one host thread per device inside the
 * call) and writes the rows into the caller's array in input order -- no collective on the data path.  The reference
 * event model crosses PCIe once, to devices[0], and travels device to device from there (hipMemcpyPeer, xGMI between the
 * GPUs of a node): the "broadcast, root 0" of the multi-GPU design.  A device may be listed more than once (two shards on
 * one GPU).  Works with sfa_align_batch, sfa_submit_batch / sfa_wait_batch, sfa_align_events, sfa_align_raw(_ex),
 * sfa_set_option (applies to every shard), sfa_sync, sfa_get_profile (slowest shard's times, summed counts), sfa_destroy;
 * sfa_align_batch_device and sfa_stream need a single-device context (device memory and streams belong to one GPU). */
int sfa_init_devices(sfa_ctx_t **ctx, const sfa_ref_t *ref, uint32_t flag, const int *devices, int n_devices);

/* Shards behind a context: 1 for sfa_init, n_devices for sfa_init_devices. */
int sfa_n_devices(sfa_ctx_t *ctx);

/* Align a batch.  queries: concatenated, already z-normalised event means in EVENT order (the library applies
 * the RNA reversal of src/sigfish.c:860-866 itself); q_off[n_reads+1] offsets into queries; a read with
 * q_off[i+1]==q_off[i] is skipped (valid=0).  out[n_reads] is written in input order.  Blocking. */
int sfa_align_batch(sfa_ctx_t *ctx, const float *queries, const int64_t *q_off, int32_t n_reads, sfa_result_t *out);

/* Same with queries and results resident in device memory (d_queries: floats in HBM, d_out: n_reads rows in
 * HBM); q_off stays a HOST array.  Work is enqueued on the context stream; returns after enqueueing unless
 * `sync` is non-zero. */
int sfa_align_batch_device(sfa_ctx_t *ctx, const float *d_queries, const int64_t *q_off, int32_t n_reads,
                           sfa_result_t *d_out, int sync);

/* The same call split in two, so that the host can load and event-detect batch i+1 while batch i is on the GPU
 * (the overlap the reference's strictly serial load -> process -> output loop, src/dtw_main.c:299-326, lacks).
 * sfa_submit_batch returns once the work is queued; `queries` must stay valid until sfa_wait_batch, which blocks,
 * fills out[n_reads] and must be called with the same n_reads.  One batch in flight per context: a second submit
 * waits for the first to finish and discards its rows (if that batch failed on the device, SFA_EKERNEL, the second
 * submit returns its error).  sfa_align_batch == submit + wait. */
int sfa_submit_batch(sfa_ctx_t *ctx, const float *queries, const int64_t *q_off, int32_t n_reads);
int sfa_wait_batch(sfa_ctx_t *ctx, sfa_result_t *out, int32_t n_reads);

/* align_db() shaped entry: per-read event tables exactly as db_t holds them (src/sigfish.h:177-178):
 * events[i] -> sfa_event_t array of read i, qstart[i]/qend[i] the window chosen by normalise_single
 * (src/sigfish.c:479-480); reads with n_events[i]==0 are skipped.  The window means are gathered out of the 24-byte event
 * records into page-locked memory (on a few threads once the batch is large) and uploaded from there. */
int sfa_align_events(sfa_ctx_t *ctx, const sfa_event_t *const *events, const int64_t *n_events,
                     const int64_t *qstart, const int64_t *qend, int32_t n_reads, sfa_result_t *out);

/* Options (all optional; rows never depend on them -- every setting is held to the same parity tests).  14 keys:
 *   planner
 *     "lane_widening"         0 = auto by batch size (default); 1 / 2 / 4 = fixed: rows per lane / w and lanes per read * w -- the
 *                             small-batch latency shapes
 *     "widen_below"           auto mode widens x4 when a batch has fewer wave-tasks per SIMD than this (default 5; a caller with
 *                             several batches in flight per device lowers it)
 *     "column_segments"       0 = auto (small batches cut every (contig,strand) sweep into up to 64 verified segments), 1 = off,
 *                             2..64 = that many
 *     "segment_warm_windows"  query lengths a segment starts early (default 4)
 *     "waves_per_simd"        1..8, occupancy target used when splitting the contig list into chunks (default 6)
 *     "min_slice_reads"       a batch whose checkpoints would not fit the budget at the shortest interval is cut into slices of at
 *                             least this many reads, run back to back (default 65536)
 *   pass 1 -> pass 2 hand-over
 *     "lds_ckpt"              1 = default: where every shape of the batch has <= 16 rows per lane (queries up to 256 events) the fill
 *                             keeps its last two snapshots in LDS and writes one to HBM only when a window becomes a read's best so
 *                             far, plus a sparse store every 32768 steps for pass 2 to back off to (--dtw-std: the sparse store
 *                             alone); 2 = the same whatever the batch size and up to 1024 events at 16 rows per lane; 0 = every
 *                             snapshot to HBM.  Shapes with 32 rows per lane always keep their snapshots in HBM.
 *     "fused_trace"           1 = default: pass 2 runs inside the fill launch (tickets) when the launch has more wave-tasks than the
 *                             device has wave slots; 2 = always; 0 = always as its own launch
 *     "ckpt_interval"         0 = auto, else a power of two >= 4: snapshots of pass 1 in HBM every that many steps
 *     "ckpt_budget_bytes"     HBM the snapshots of one batch may take (default 32 GiB)
 *     "trace_margin"          -1 = auto: pass 2 starts a query length (+ lanes) in front of the winning window -- on the HBM-snapshot
 *                             route as far as 99.9 % of the PREVIOUS batch's alignments spanned, rounded up to the next sixteenth of
 *                             the query, + 1/16 query + lanes + 16; a read whose path is longer backs off one snapshot; >= 0: that
 *                             many steps
 *   launch
 *     "prio_unit"             columns per level of the fill's longest-remaining-first issue priority in the tail of a launch
 *                             (default 2048; 0 = off)
 *     "spin_limit_ms"         default 20000: the longest a wave of a launch waits for another wave of the same launch -- pass 2 for
 *                             its quad's fill tasks, a row strip for the strip above -- before the batch fails with SFA_EKERNEL;
 *                             floors apply (about five times the longest fill task; the strips' pipeline depth)
 *   raw-signal path
 *     "ev_parallel"           bit 0: wave-per-read prefix sums for every read whose sums are provably exact in any order (the
 *                             sequential kernel for the rest); bit 1: chunk-parallel peak picker accepted where it is certified
 *                             to equal the sequential one; default 3
 * Test hooks, refused unless SFA_TEST_HOOKS=1 is in the environment: "debug_drop_quad" / "debug_drop_strip" (the producer with
 * this index never signals; every batch then fails with SFA_EKERNEL after the wait limit; -1 = off). */
int sfa_set_option(sfa_ctx_t *ctx, const char *key, int64_t value);

/* Plan a batch without running it: how reads would be grouped.  slot_of_read[n_reads] (may be NULL) receives
 * quad*4+slot per read or -1 for skipped reads; job_len[n_jobs] are the (contig,strand) lengths in processing
 * order.  ckpt_interval / ckpt_budget_bytes / lane_widening as in sfa_set_option (0 = defaults; the device is
 * assumed to have 1024 SIMDs). */
int sfa_plan_batch(const int64_t *q_off, int32_t n_reads, const int32_t *job_len, int32_t n_jobs, int64_t ckpt_interval,
                   int64_t ckpt_budget_bytes, int32_t lane_widening, int32_t *slot_of_read, sfa_plan_info_t *info);

/* Block until everything enqueued on the context stream has finished. */
int sfa_sync(sfa_ctx_t *ctx);

/* Timing of the most recent align call (valid after it completed / after sfa_sync). */
int sfa_get_profile(sfa_ctx_t *ctx, sfa_profile_t *prof);

/* The HIP stream (hipStream_t) the context enqueues on, as an opaque pointer. */
void *sfa_stream(sfa_ctx_t *ctx);

void sfa_destroy(sfa_ctx_t *ctx);

const char *sfa_last_error(void);
const char *sfa_version(void);
/* Identity of the device code this library was built from (hash of the kernel and launch sources): profiles/ files are
 * stamped with it, so that numbers measured on one build are never quoted for another. */
const char *sfa_build_id(void);

/* ---- host-side helpers on the same path (no GPU needed) ------------------------------------------------ */

/* Reference event model from sequences (gen_ref, src/genref.c:86-241), one record at a time.
 * level_mean[4^k]: k-mer model means.  fwd/rev must hold (len+1-k) floats (rev may be NULL for RNA).
 * Returns the k-mer count ref_len (or <0 on error) and stores ref_st_offset. */
int32_t sfa_gen_ref_record(const char *seq, int32_t len, const float *level_mean, uint32_t k, uint32_t flag,
                           int32_t query_size, float *fwd, float *rev, int32_t *st_offset);

/* z-normalisation used for both queries and reference arrays (src/sigfish.c:483-502, src/genref.c:23-47). */
void sfa_znormalise(float *v, uint64_t n);

/* One PAF line for a result row (paf_str, src/sigfish.c:628-660).  Returns bytes written (excluding NUL),
 * or a negative value if cap is too small. */
int sfa_paf_row(char *buf, size_t cap, const sfa_result_t *r, const char *read_id, const char *rname,
                uint64_t start_raw_idx, uint64_t end_raw_idx, uint64_t query_size, uint64_t len_raw_signal,
                uint64_t rlength);

/* ---- raw signal in, result rows out: the pre-DP stages on the GPU as well --------------------------------- */

/* What the output writer needs besides the result row (aln_to_str, src/sigfish.c:796-826). */
typedef struct {
    int64_t n_events;       /* events detected in the read (0: read dropped) */
    int64_t qstart, qend;   /* query window in events (src/sigfish.c:479-480) */
    uint64_t start_raw_idx; /* event[qstart].start */
    uint64_t end_raw_idx;   /* event[qend-1].start + length */
    int32_t status;         /* bit 0: too short (kept), bit 1: ignored (dropped) */
    int32_t pad;
} sfa_query_info_t;

/* process_db() for a whole batch on the device (src/sigfish.c:1018-1047 minus parsing): raw ADC samples -> pA ->
 * event detection -> query window -> z-normalisation -> alignment.  raw: concatenated int16 samples, raw_off[n+1];
 * scaling[3*i..]: digitisation, offset, range of read i (slow5 record fields).  prefix_size must be >= 0 (the RNA
 * "-p -1" adaptor/poly-A detection stays on the host path: sfa_detect_events + sfa_select_query).
 * rows[n] and info[n] are written in input order.  Blocking. */
int sfa_align_raw(sfa_ctx_t *ctx, const int16_t *raw, const int64_t *raw_off, const double *scaling, int32_t n_reads,
                  int32_t prefix_size, int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info);

/* Same, and the event table of every read's query window comes back as well: query_events[i * query_size + e], e <
 * info[i].qend - info[i].qstart, holds event qstart + e of read i with its mean z-normalised -- what sfa_sam_row needs
 * (pass it with qstart = 0, qend = the window length).  query_events may be NULL (then identical to sfa_align_raw). */
int sfa_align_raw_ex(sfa_ctx_t *ctx, const int16_t *raw, const int64_t *raw_off, const double *scaling, int32_t n_reads,
                  int32_t prefix_size, int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info, sfa_event_t *query_events);

/* ---- BLOW5 records in, result rows out: the record decoder on the GPU as well ------------------------------ */

/* What the output writer needs from a record besides its alignment (slow5_rec_t fields, slow5lib/include/slow5/slow5.h). */
typedef struct {
    char read_id[128];     /* NUL terminated */
    int32_t id_len;
    int32_t pad;
    int64_t n_samples;     /* len_raw_signal */
    double digitisation, offset, range;
    int64_t record_bytes;  /* on-disk size of the record (the reference's -B accounting, src/sigfish.c:304) */
} sfa_read_head_t;

/* load_db()'s records straight to the device (src/sigfish.c:274-314 hands them to parse_single, 317-328, on host threads):
 * records: the BLOW5 records of a batch back to back WITHOUT their u64 size prefixes, rec_off[n+1] their offsets;
 * record_zlib / signal_svb: the file's compression methods (record_press == zlib, signal_press == svb-zd).  Every record
 * is inflated (own DEFLATE decoder, one lane per record), its primary fields are parsed and its signal decoded
 * (StreamVByte zig-zag deltas, one wave per record) on the device; the path of sfa_align_raw_ex continues from there.
 * heads[n] receives the fields the output needs.  A record the device decoder declines (malformed, longer read id than
 * sfa_read_head_t holds, inflating to more than 4x its size + 4 KB) sends the batch through the library's host reader
 * instead -- same results, counted in sfa_profile_t.blow5_fallbacks.  Other arguments as for sfa_align_raw_ex. */
int sfa_align_blow5(sfa_ctx_t *ctx, const uint8_t *records, const int64_t *rec_off, int32_t n_reads, int32_t record_zlib,
                    int32_t signal_svb, int32_t prefix_size, int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info,
                    sfa_read_head_t *heads, sfa_event_t *query_events);

/* The device-side DEFLATE decoder on its own (tests): n zlib streams in[in_off[i]..in_off[i+1]) -> out[out_off[i]..),
 * out_len[i] = bytes produced or -1 (malformed, or larger than its slot).  Single-device context. */
int sfa_inflate_zlib_device(sfa_ctx_t *ctx, const uint8_t *in, const int64_t *in_off, int32_t n, uint8_t *out, const int64_t *out_off,
                            int32_t *out_len);

/* Page-locked host memory for the buffers handed to sfa_align_raw / sfa_align_batch (uploads from pageable memory
 * run at a fraction of the PCIe rate).  Plain malloc-style pair; NULL on failure. */
void *sfa_pinned_alloc(size_t bytes);
void sfa_pinned_free(void *p);

/* ---- host pre-DP stages and readers (no GPU needed) ----------------------------- */

/* event_single (src/sigfish.c:330-378): raw ADC samples -> pA -> events (getevents, src/events.c:557-577).
 * Writes at most cap events to out; returns the number of events detected (call again with a larger buffer if
 * the return value exceeds cap), or <0 on error. */
int64_t sfa_detect_events(const int16_t *raw, int64_t n_raw, double digitisation, double offset, double range, int rna,
                          sfa_event_t *out, int64_t cap);

/* normalise_single (src/sigfish.c:424-505): choose the query window [qstart,qend) from -p/-q/--from-end
 * (prefix_size < 0: RNA adaptor/poly-A auto detection, src/sigfish.c:380-422 + src/jnn.c) and z-normalise those
 * event means in place.  Returns 1 if the read is kept, 0 if the reference would drop it (et.n = 0). */
int sfa_select_query(sfa_event_t *events, int64_t n_events, const int16_t *raw, int64_t n_raw, double digitisation,
                     double offset, double range, int32_t prefix_size, int32_t query_size, uint32_t flag, int pore,
                     int64_t *qstart, int64_t *qend);

/* One SAM line for a result row (sam_str, src/sigfish.c:770-794, with path_to_map 530-571 and the "ss" string of
 * r2qevent_map_to_ss 663-768).  The warp path of the winner is rebuilt on the host from the band between its
 * start and end columns.  events/qstart/qend: the read's event table and query window as for sfa_align_events
 * (means z-normalised); ref_array: the winner's (contig,strand) array -- forward[rid] for '+', reverse[rid] for
 * '-' -- of ref_len floats; ref_st_offset as given to sfa_init.  Returns bytes written or <0. */
int sfa_sam_row(char *buf, size_t cap, const sfa_result_t *r, const char *read_id, const char *rname,
                const sfa_event_t *events, int64_t qstart, int64_t qend, const float *ref_array, int32_t ref_len,
                int32_t ref_st_offset, uint32_t flag);

/* aln_t.r2qevent_map for a result row (path_to_map, src/sigfish.c:530-571, as update_aln stores it at 610-613): what the
 * reference's own sam_str / r2qevent_map_to_ss (src/sigfish.c:663-794) consume.  The winner's warp path is rebuilt on the
 * host from the band between its start and end columns, exactly as for sfa_sam_row; arguments as there.
 * pairs[2*i], pairs[2*i+1] = start, stop (query event indices relative to qstart, in DP order) for reference column
 * pos_st + i -- the memory layout of index_pair_t[] (src/sigfish.h:141-144), so a C host may pass its own array.
 * Returns r2qevent_size = pos_end - pos_st + 1 (also when pairs is NULL: size query), SFA_ERANGE if cap_pairs (in pairs)
 * is smaller than that, SFA_EINVAL for an unaligned row. */
int32_t sfa_r2qevent_map(const sfa_result_t *r, const sfa_event_t *events, int64_t qstart, int64_t qend, const float *ref_array,
                         int32_t ref_len, int32_t ref_st_offset, uint32_t flag, int32_t *pairs, int32_t cap_pairs);

/* read_model (src/model.c:38-131): text k-mer model -> level_mean[4^k] (levels must hold 262144 floats).  Accepts and refuses what
 * the reference does; rows it would only have logged as corrupted (src/model.c:98-100) are counted like it counts them and listed
 * in sfa_last_error() after a successful return (empty string: none). */
int sfa_read_kmer_model(const char *path, float *levels, uint32_t *k);

/* Sequential S/BLOW5 reader, the subset of slow5lib the path uses (slow5_open, src/sigfish.c:110): BLOW5 (zlib / svb-zd /
 * uncompressed) and its text twin SLOW5 ASCII, told apart by content (slow5lib: by the extension).  Numbers in a text file are
 * accepted as slow5lib accepts them (slow5_misc.c:103-156); auxiliary columns are counted, not interpreted. */
typedef struct sfa_blow5 sfa_blow5_t;
sfa_blow5_t *sfa_blow5_open(const char *path);            /* NULL on failure, see sfa_last_error() */
const char *sfa_blow5_attr(sfa_blow5_t *f, const char *key); /* header attribute of read group 0 or NULL */
/* 1: a record was read, 0: end of file, <0: error.  Pointers stay valid until the next call on f.
 * meta = {digitisation, offset, range, sampling_rate}. */
int sfa_blow5_next(sfa_blow5_t *f, const char **read_id, double meta[4], const int16_t **raw, int64_t *n_raw);
void sfa_blow5_close(sfa_blow5_t *f);
/* One rank's part of a read-sharded run (replaces the whole-file loop of src/dtw_main.c:299-326 by G loops over disjoint parts;
 * records are framed by their u64 size prefixes only, slow5lib/src/slow5.c:3218-3266, so a part is found by walking them).
 * Call after sfa_blow5_open and before the first sfa_blow5_next; past the selection sfa_blow5_next returns 0.
 *   sfa_blow5_select_shard    the records whose size prefix starts in the r-th of G equal byte slices of the record region:
 *                             the G shards are every record exactly once, in file order (regular files only)
 *   sfa_blow5_select_records  records [first, first + count) by position in the file (count < 0: to the end) */
int sfa_blow5_select_shard(sfa_blow5_t *f, int32_t r, int32_t G);
int sfa_blow5_select_records(sfa_blow5_t *f, int64_t first, int64_t count);

/* Free and total memory of a device in bytes (hipMemGetInfo), for callers sizing their batches. */
int sfa_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* The record decompressor of the BLOW5 reader on its own: inflates the zlib stream in[0..n) into out[0..cap) with the
 * library's own DEFLATE decoder and returns the number of bytes produced, SFA_ERANGE when cap is too small, SFA_EINVAL
 * when the decoder declines the stream (the reader then falls back to zlib's inflate). */
int64_t sfa_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap);

/* The same for TWO streams decoded side by side by the calling thread (how the reader's host threads take records: the symbol
 * loops of the two streams take turns, two dependence chains keep a core busier than one).  len[k] receives what
 * sfa_inflate_zlib would return for stream k -- a stream that is declined does not disturb the other.  Returns SFA_OK, or
 * SFA_EINVAL for null arguments. */
int sfa_inflate_zlib_pair(const uint8_t *in0, size_t n0, const uint8_t *in1, size_t n1, uint8_t *out0, size_t cap0, uint8_t *out1,
                          size_t cap1, int64_t len[2]);

#ifdef __cplusplus
}
#endif
#endif /* SIGFISH_AMD_H */
