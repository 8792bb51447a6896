// sdtw_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels for the subsequence-DTW alignment stage.
//
// What is computed (reference: hasindu2008/sigfish v0.2.0):
//   * accumulated-cost recurrence of subsequence() / std_dtw()      src/cdtw.c:171-189 / 69-94
//   * the alignment start column that subsequence_path() would       src/cdtw.c:98-167, 192-227
//     trace back to, propagated FORWARD with the same tie order
//     (diagonal, then left, then up), so no matrix is ever stored
//   * the per-window first-strict-minimum scan of the last row       src/sigfish.c:891-901, 938-948
//   * the top-2 of the reference's sorted top-5 candidate list       src/sigfish.c:575-626 (ties: later wins)
//   * strand flip, ref_st_offset, mapq                               src/sigfish.c:969-983
//
// Mapping to the machine (MI355X-first, not a translation of the CPU loops):
//   * one read occupies ONE DPP ROW (16 lanes) of a wave64; four reads ("a quad": one length class, lengths equal modulo the
//     rows per lane so that every read's last query row sits in the same lane and register -- MixedQuad) ride in
//     a wave.  Lane g of a row owns R consecutive query rows (R = 4/8/16/32 -> queries up to 64/128/256/512
//     events), kept in VGPRs together with their running cost.  No barriers, no cost matrix.  Queries longer
//     than 512 events take 32 or 64 lanes per read at R = 32 (two reads / one read per wave, up to 1024 / 2048
//     events); everything below is written for L lanes per read.  SMALL BATCHES use the same freedom the other
//     way round: when there are too few reads to fill the chip the planner quarters R and quadruples L
//     (q = 250: 4 rows x 64 lanes; halving / doubling is an option), which gives 4x the waves, each with a 4x
//     shorter step -- latency per batch drops accordingly.
//   * the lanes of a row walk an anti-diagonal: at step t lane g is at reference column t-g, so the only
//     cross-lane traffic per step is ONE value per lane, the bottom cost handed to the next lane (through a
//     wave-private LDS window, see Exchange): the neighbour's value from the previous step is this lane's "up",
//     the one before its "diagonal".
//   * +inf initial state makes not-yet-started columns (t-g < 0) and past-the-end columns harmless, so the inner
//     loop carries no per-lane predication; reference arrays are padded in HBM so the per-lane 16-byte loads of
//     four upcoming reference levels never leave the allocation (they hit L1/L2: every wave streams the same
//     few hundred KB).
//   * TWO PASSES.  Pass 1 (sdtw_fill_kernel, >95 % of the time) evaluates costs only -- v_sub, v_min3,
//     v_add(|d|) per cell -- finds every window minimum and the per-read top-2, and drops a checkpoint of the
//     (R+1)-register anti-diagonal state (plus the step index) every T steps.  Pass 2 (sdtw_trace_kernel) re-runs, for each read's
//     WINNING candidate only, the few hundred steps between a checkpoint and the winning cell with start-column
//     tracking switched on (+2 v_cmp_eq, +2 v_cndmask per cell); if the path turns out to begin before the
//     checkpoint (start sentinel -1 survives) it backs off to an earlier one, ultimately to step 0.  Costs are
//     restored bit-exactly from the checkpoint, so the recovered start equals the full traceback's.
//   * all arithmetic is IEEE fp32 with denormals, no FMA contraction: every cell is bit-identical to the
//     reference's row-major evaluation because each cell is a pure function of its three neighbours.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfa {

constexpr int kRefPad = 128;       // floats of +inf padding on both sides of every (contig,strand) array in HBM (>= 64 lanes + 2 loads)
#ifndef SFA_STEPS_PER_LOAD
#define SFA_STEPS_PER_LOAD 4
#endif
#ifndef SFA_FILL_WAVES
#define SFA_FILL_WAVES 6  // waves per SIMD the cost-only fill (R <= 16) is register-budgeted for: 80 VGPRs
#endif
#ifndef SFA_FILL32_WAVES
#define SFA_FILL32_WAVES 4  // the R = 32 shapes: 128 VGPRs (a handful of spills outside the loop); 81.5 -> 79.6 ms at q = 500
#endif
#ifndef SFA_LCK_WAVES
#define SFA_LCK_WAVES 4  // the fill with its checkpoints in LDS: 2 x 17 planes x 256 B per wave -> four blocks of four waves per CU
#endif
#ifndef SFA_TRACE_WAVES
#define SFA_TRACE_WAVES 4  // waves per SIMD pass 2 (R <= 16) is register-budgeted for: 128 VGPRs; 3.49 -> 3.28 ms per 100 k reads against 144 VGPRs / 3 waves
#endif
constexpr int kStepsPerLoad = SFA_STEPS_PER_LOAD;  // reference levels fetched per load (4 = one 16-byte load)
constexpr int kSpanBuckets = 32;   // histogram of alignment spans in sixteenths of the query length (FinalizeArgs::span_hist)
constexpr int kMaxClasses = 6;     // query-length classes; base shapes (R, lanes) = (32,64) (32,32) (32,16) (16,16) (8,16) (4,16)

struct __attribute__((packed, aligned(4))) float4u {
    float v[kStepsPerLoad];
};

// A query-length class inside one launch.
struct ClassDesc {
    int32_t R;          // query rows per lane
    int32_t lanes;      // lanes per read: 16 (four reads per wave), 32 (two) or 64 (one)
    int32_t quad_base;  // first quad (a "quad" is one wave's worth of reads: 64/lanes of them, slots 0..)
    int32_t n_quads;
    int32_t task_base;  // first task (fill: n_quads*n_chunks tasks, trace: n_quads tasks)
    int64_t ck_base;    // float offset of this class's checkpoint region
};

// Everything the fill / trace kernels read.  Passed by value (kernarg segment).
struct DpArgs {
    const float *queries;        // HBM: concatenated z-normalised event means, event order
    const int64_t *q_off;        // [n_reads+1]
    const int32_t *order;        // [n_quads_total*4] read index per (quad, slot) or -1
    const int32_t *quad_qlen;    // [n_quads_total] query length shared by the quad's reads
    const float *ref;            // padded reference event arrays
    const int64_t *job_off;      // [n_jobs] offset of column 0 of job (contig,strand) in `ref`
    const int32_t *job_len;      // [n_jobs] rlen
    const int32_t *chunk_begin;  // [n_chunks+1] job ranges
    const int32_t *job_ck_off;   // [n_jobs+1] prefix sum of checkpoints per job (for the current interval)
    float *ck;                   // checkpoint store
    // partial results of the fill, index (quad*n_chunks+chunk)*4+slot
    float *p_best;
    int32_t *p_end;
    int32_t *p_job;
    float *p_second;
    // winners per read (written by finalize, read by trace)
    int32_t *w_job;
    int32_t *w_end;    // first column of the winning WINDOW (the trace kernel finds the cell inside it); 32-row shapes: the winning cell's column
    float *w_score;    // the winning score, to recognise that cell
    ClassDesc cls[kMaxClasses];
    int32_t n_cls;
    int32_t n_chunks;
    int32_t n_tasks;
    int32_t rev_query;     // 1: query rows are the events reversed (RNA without --invert)
    int32_t ck_shift;      // checkpoint interval T = 1 << ck_shift steps; 0 = no checkpoints
    int32_t trace_margin;  // pass 2 starts from the last checkpoint at least this many steps before the winner
    int32_t n_reads_total; // reads in the batch (trace output: start columns [n], end columns [n])
    // column segments (small batches, 64-lane shapes, cost-only sDTW): a chunk is (job, segment), see sweep_segment()
    int32_t n_seg;         // segments per job (1 = off)
    int32_t warm_windows;  // query-length windows a segment starts before its own first window
    int32_t n_jobs;
    int32_t verify_planes; // floats per lane in a hand-over snapshot (max R + 1)
    float *verify;         // [quad][job][segment][in,out][verify_planes][64]
    int32_t *seg_fail;     // [quad] set when a hand-over does not match
    // rolling checkpoints in LDS (LCK kernels, see LdsCkpt): the record of every (quad, chunk)'s current best window,
    // [task][R_max + 1 planes][64 lanes]; the step index of that snapshot per (task, slot) (-1: none, start from scratch);
    // the best score of every read so far over all its tasks (float bits, for the save filter); the winning chunk per read
    float *best_rec;
    int32_t *best_e;
    unsigned *g_best;
    int32_t *w_chunk;
    int32_t best_planes;   // planes per record (R of the largest class + 1)
    int32_t coarse_every;  // every coarse_every-th LDS snapshot also goes to the HBM checkpoint store (ck_shift = lck_shift + log2 of it)
    int32_t lck_shift;     // LDS snapshots every 1 << lck_shift steps (9, 10 or 11 by the longest query of the batch)
    // fused launch (FUSED kernels): waves claim tickets; tickets < n_tasks are fill tasks, ticket n_tasks + q is pass 2 of quad q,
    // which waits until quad_done[q] == n_chunks.  Rows are written by the pass-2 waves (tables as FinalizeArgs).
    const struct DpArgs *self;  // this very block in device memory: what the pass-2 function of the fused launch reads (see there)
    unsigned *ticket;
    int32_t *quad_done;
    int32_t n_quads_total;
    const int32_t *job_contig;
    const int8_t *job_strand;
    const int32_t *ref_len;
    const int32_t *ref_st_offset;
    const uint8_t *bad;
    struct ResultRow *out;
    unsigned *span_hist;   // fused launch on the 32-row fill: spans of this batch's alignments (see FinalizeArgs::span_hist); else nullptr
    // longest-remaining-first issue priority in the tail of the launch (see IssuePriority): columns per priority step,
    // 0 = off; `started` counts the tasks that have begun (zeroed before the launch)
    int32_t prio_unit;
    unsigned *started;
    // bounded waits (see bounded_wait_ge): error words of the launch, limit in 100 MHz ticks; debug_drop_quad >= 0: the fill
    // tasks of that quad do not count themselves in (a never-published quad, for the test of the bound)
    unsigned *err;
    long long spin_limit;
    int32_t debug_drop_quad;
    unsigned long long *task_times;  // -DSFA_TASK_TIMES builds only (tools/task_times.py): [task][3] start, end (100 MHz ticks), SIMD position
};

// physical position of the executing wave: (xcc, se, sh, cu, simd) from HW_REG_XCC_ID / HW_REG_HW_ID, 14 bits.  Measured
// on MI355X (tools/hwid_probe.hip): a grid of 1536 four-wave blocks covers 1024 distinct positions with 6 waves each.
constexpr int kSimdSlots = 16384;
__device__ __forceinline__ unsigned simd_position() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID[3:0]
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return ((((xcc & 15) * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd;
}

// ---- hand-over between waves of ONE launch (fused pass 2, pipelined row strips) ----------------------------------------
// Producer: every byte the consumer will read is stored WRITE-THROUGH (global_store ... sc1: relaxed agent-scope atomic
// stores, or inline asm for the 16-byte rows), then the storing wave waits for its own stores -- drain_stores() -- and only
// then bumps the counter / progress word the consumer polls.  The wait is INLINE ASM on purpose: a workgroup-scope release
// fence emits no vmcnt wait at all on gfx950 / ROCm 7.2 (round 2 relied on it: the counter could overtake the stores), and
// the compiler may drop the wait of an agent-scope fence when it believes the scoreboard empty; the asm statement is
// invisible to that pass and a compiler barrier for memory operations.  tests/test_publish_isa.py disassembles the shipped
// kernels and checks stores (sc1), wait and counter are there in this order.  The agent-scope release fence itself
// (buffer_wbl2) is what this avoids: it writes the whole L2 of the XCD back at the end of every task.
// Consumer: relaxed poll of the counter, agent-scope acquire (buffer_inv sc1), then plain loads.
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// One write-through store with its offset as an immediate.  The R + 2 planes of a checkpoint record are 256 B apart; written as
// 34 relaxed atomic stores the compiler materialises 34 64-bit addresses up front and the 32-row fill (128 VGPRs, no slack) spills
// 1 100 registers into its step loops; with immediate offsets off three base pointers the record costs 6 address registers.
template <int OFF>
__device__ __forceinline__ void store_wt(const float *base, const float v) {
    static_assert(OFF >= 0 && OFF < 4096, "12-bit immediate");
    asm volatile("global_store_dword %0, %1, off offset:%2 sc1" ::"v"(base), "v"(v), "n"(OFF) : "memory");
}
// planes [R costs | dprev | steps elapsed] of a checkpoint record, write-through (read by pass-2 waves of the same launch)
template <int R, int I = 0, typename CV>
__device__ __forceinline__ void store_record_wt(float *ckp, const CV &cv, const float dprev, const int e) {
    if constexpr (I < R + 2) {
        const float v = I < R ? static_cast<float>(cv[I < R ? I : 0]) : (I == R ? dprev : __int_as_float(e));
        store_wt<(I % 16) * 256>(ckp + (I / 16) * 16 * 64, v);
        store_record_wt<R, I + 1>(ckp, cv, dprev, e);
    }
}

// A wait that never ends is a hung stream and, on this pool, a lost box.  Every spin of a launch is bounded by wall-clock
// time (s_memrealtime: 100 MHz): when the limit passes the wave records what it waited for in the launch's error words,
// gives up, and the host returns SFA_EKERNEL for the batch (rows are not to be used).  err[0] = code (first error wins),
// err[1], err[2] = detail.
constexpr unsigned kErrQuadWait = 1;   // fused launch: pass 2 of quad err[1] never saw its fill tasks complete (err[2] = count seen)
constexpr unsigned kErrStripWait = 2;  // pipelined strips: a strip never saw the row above reach column err[1] (err[2] = column seen)
__device__ __forceinline__ void report_device_error(unsigned *err, unsigned code, int d1, int d2) {
    if (atomicCAS(err, 0u, code) == 0u) {
        (void)__hip_atomic_exchange(err + 1, static_cast<unsigned>(d1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_exchange(err + 2, static_cast<unsigned>(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// lane 0 polls *ctr until it is >= need; returns false (all lanes) when `limit` ticks have passed.  sleep: s_sleep argument
__device__ __forceinline__ bool bounded_wait_ge(const int32_t *ctr, const int need, const long long limit, int *seen, const bool long_sleep) {
    int ok = 1, v = 0;
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t0 = wall_clock64();
        while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
            if (long_sleep)
                __builtin_amdgcn_s_sleep(127);
            else
                __builtin_amdgcn_s_sleep(16);
            if (static_cast<long long>(wall_clock64() - t0) > limit) {
                ok = 0;
                break;
            }
        }
    }
    *seen = __builtin_amdgcn_readfirstlane(v);
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// Neighbour exchange.  Lane g needs the bottom cost lane g-1 produced in the previous step.  On gfx950 a
// `v_mov_b32_dpp row_shr:1` in this dependent position costs the SIMD ~40 issue cycles per step (measured,
// tools/valu_ceiling.hip: 0.34 vs 0.45 VALU instructions per cycle per SIMD), so the value makes a round trip
// through a wave-private LDS window instead: every lane stores its bottom value in word g+1 of its read's
// window and loads word g; word 0 holds the boundary value for query row 0.  LDS operations of one
// wave execute in order, the windows are private to the wave, hence no barrier; the LDS pipe is otherwise idle.
// Bank-conflict-free layout (ds_*_b32: 32 banks, two 32-lane groups), L = lanes per read: read r's window is the
// L words [L*r, L*r+L): word 0 = boundary, lane g (< L-1) stores to word g+1, every lane g loads word g.  The last
// lane's value has no reader; it goes to a private dummy word at 64+L*r (the banks its 32-lane group leaves free).
constexpr int kXchWordsPerWave = 128;

struct Exchange {
    float *wf, *rf;  // this lane's store / load slot (bottom cost)
    int *wi, *ri;    // same for the bottom start column, TRACK only
    // g0: the lane that holds query row 0 of this read (0 unless the read shares its wave with longer ones, see MixedQuad): that
    // lane loads the boundary word instead of its neighbour's value, so the lanes in front of it feed nothing
    __device__ __forceinline__ void init(float *lds_f, int *lds_i, int wave_in_block, int slot, int g, int L, int g0 = 0) {
        const int base = wave_in_block * kXchWordsPerWave;
        const int w = base + (g < L - 1 ? slot * L + g + 1 : 64 + L * slot);
        const int r = base + slot * L + (g == g0 ? 0 : g);
        wf = lds_f + w;
        rf = lds_f + r;
        wi = lds_i + w;
        ri = lds_i + r;
    }
    // boundary seen by lane 0 (query row 0): 0 = free start of subsequence(); std_dtw() switches it to +inf after column 0
    template <bool TRACK>
    __device__ __forceinline__ void set_boundary(bool lane0, float vf) {
        if (lane0) {
            *((lds_vf *)rf) = vf;
            if (TRACK) *((lds_vi *)ri) = 0;
        }
    }
    // volatile: the neighbour's slot is written by ANOTHER lane, which the single-thread memory model cannot see;
    // the explicit LDS address space keeps these as plain ds_write_b32 / ds_read_b32
    typedef __attribute__((address_space(3))) volatile float lds_vf;
    typedef __attribute__((address_space(3))) volatile int lds_vi;
    __device__ __forceinline__ float shift(float bottom) {
        *((lds_vf *)wf) = bottom;
        return *((lds_vf *)rf);
    }
    __device__ __forceinline__ int shift(int bottom) {
        *((lds_vi *)wi) = bottom;
        return *((lds_vi *)ri);
    }
};

// Per-lane state lives in R-wide register tuples.  A wave-uniform (SGPR) index into such a tuple lowers to
// s_set_gpr_idx_on / v_mov_b32 / s_set_gpr_idx_off on gfx950: one VALU op to read "the register that holds the
// last query row", whatever the query length.
template <typename T, int R>
struct Vec {
    typedef T type __attribute__((ext_vector_type(R)));
};

// Shapes with 32 rows per lane (queries of 257 .. 2048 events): the cost-only pass 1 names the winning CELL -- the column of the
// first strict minimum of every window (src/sigfish.c:892-899), two operations per step of ~100 -- instead of the window alone,
// so that pass 2 starts a query length in front of that cell rather than in front of its window: a quarter to a third of its
// steps.  (The 16-row shapes cannot afford two more operations on 49; their pass 2 keeps scanning the window.)
template <int R, bool STD>
struct CellFromFill {
    static constexpr bool value = !STD && R >= 32;
};

// Running top-2 of the reference's candidate list for one read (kept in the registers of the lane that owns
// the last query row).  Insertion rule of update_aln (src/sigfish.c:577-583): a candidate goes in front of
// everything that is not strictly better, so on equal scores the LATER candidate ranks higher.
struct Top2 {
    float best, second;
    int32_t end, job;
    __device__ __forceinline__ void init() {
        best = INFINITY;
        second = INFINITY;
        end = -1;
        job = -1;
    }
    __device__ __forceinline__ bool offer(float sc, int32_t pos, int32_t j) {
        const bool top = !(sc > best);
        const bool sec = !(sc > second);
        second = top ? best : (sec ? sc : second);
        best = top ? sc : best;
        end = top ? pos : end;
        job = top ? j : job;
        return top;
    }
    // the same for the lanes where `on` holds (reads of a wave whose windows end at different columns)
    __device__ __forceinline__ bool offer_if(bool on, float sc, int32_t pos, int32_t j) {
        const bool top = on && !(sc > best);
        const bool sec = on && !(sc > second);
        second = top ? best : (sec ? sc : second);
        best = top ? sc : best;
        end = top ? pos : end;
        job = top ? j : job;
        return top;
    }
};

// MIXED QUADS.  The reads of a wave used to have ONE query length (the planner grouped by length), so a ragged batch -- a few
// hundred distinct lengths -- left most of the short reads' waves a quarter or half full.  Reads of different lengths can share
// a wave when they are laid out from the END: every read's LAST query row sits in the same lane and register (lq, rq: those of
// the longest read of the wave, whose length is quad_qlen), a shorter read simply begins in a later lane g0 -- at register 0 of
// that lane, which is why the lengths of a wave agree modulo R (the planner's rule).  Lane g0 takes the boundary of query row 0
// (Exchange::init), the lanes in front of it compute cells nobody reads.  What differs per read is the WINDOW length of the
// last-row scan (src/sigfish.c:891-901: windows of qlen columns), handled in sweep_job; nothing in the step itself.
struct MixedQuad {
    int myq;  // query length of this lane's read (the wave's longest for an empty slot)
    int g0;   // first lane of the read's L-lane row that holds query rows
    template <int R>
    __device__ __forceinline__ void init(const DpArgs &a, int read, int qmax) {
        myq = read >= 0 ? static_cast<int>(a.q_off[read + 1] - a.q_off[read]) : qmax;
        g0 = (qmax - myq) / R;
    }
};

// Issue priority by remaining work, in the tail of a launch.  The SIMD's arbiter serves the OLDEST ready wave first, so
// six waves that start together do not advance together: measured (tools/task_times.py, 6 equal tasks per SIMD) they end
// at 4.5, 4.8, 6.7, 8.3, 9.9 and 11.2 ms -- the last ones run alone, at the 40 % of the SIMD's rate that one wave's
// dependent chain can use, and the batch takes 11.2 ms where its share of a large one is 9.3.  s_setprio outranks age: a
// wave lowers its own priority as its remaining columns fall below 4U, 2U, U, so whoever has the most left on a SIMD goes
// first and the waves of a SIMD converge on a common finish (same test: 9.2 ... 9.7 ms).  While tasks are still waiting
// for a slot that is the wrong policy -- everything on a SIMD would finish at once and its successors start late (measured:
// +3 % on 100 000 reads) -- so it only applies once the LAST task of the launch has started: every wave counts itself in
// `started` when it begins and looks at the counter every few windows until it reads n_tasks.
__device__ __forceinline__ void set_issue_priority(int level) {
    switch (level) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
struct IssuePriority {
    int unit;       // U; 0 = off
    int remaining;  // columns of the task still ahead when the current job starts (this job included)
    int next_drop;  // `remaining - col` value at which the level falls next
    int countdown;  // windows until the next look at the counter; < 0: the tail has begun
    const unsigned *started;
    unsigned n_tasks;
    __device__ __forceinline__ void start(int u, int rem, unsigned *started_ctr, unsigned total, bool count_me = true) {
        unit = u;
        remaining = rem;
        next_drop = 0x7fffffff;
        countdown = 1;
        started = started_ctr;
        n_tasks = total;
        if (unit > 0) {
            // top level until the tail is seen: a wave left at the default (lowest) level next to waves that already lowered
            // theirs step by step from 3 would starve, never reach its next look at the counter, and finish alone
            set_issue_priority(3);
            if (count_me && (threadIdx.x & 63) == 0) __hip_atomic_fetch_add(started_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __device__ __forceinline__ void update(int col) {  // at a window boundary, col columns into the current job
        const int left = remaining - col;
        const int level = left > 4 * unit ? 3 : (left > 2 * unit ? 2 : (left > unit ? 1 : 0));
        set_issue_priority(level);
        next_drop = level == 3 ? 4 * unit : (level == 2 ? 2 * unit : (level == 1 ? unit : -1));
    }
    __device__ __forceinline__ void at_window(int col) {
        if (unit <= 0) return;
        if (countdown >= 0) {
            if (--countdown > 0) return;
            countdown = 4;
            const unsigned s = __builtin_amdgcn_readfirstlane(__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (s < n_tasks) return;
            countdown = -1;
            update(col);
        } else if (remaining - col <= next_drop) {
            update(col);
        }
    }
};

// One anti-diagonal step for the R rows of this lane at reference level yv.
//   c[r]   cost of row r at this lane's previous column ("left"); updated in place
//   s[r]   start column carried with it (TRACK only)
//   dprev  the "up" input of the previous step, i.e. this step's diagonal input for row 0
//   t      step index (wave-uniform int in the fill, per-lane in the trace); lane 0's column is t
//   T0     std_dtw only: t may be 0 in this step (the special case of row 0's first column).  The cost-only fill knows that only
//          the first block of four steps of a job can hold t = 0 and runs every later block without the test (one scalar
//          compare + branch and one v_cndmask per step: 3.14 -> 3.09 VALU instructions per cell)
//   x      the lane's R query rows: a register array, or LdsRows (pass 2 of the 32-row shapes inside the fill launch)
template <int R, bool TRACK, bool STD, typename TT, bool T0 = true, typename CF, typename CI, typename XT>
__device__ __forceinline__ void dp_step(CF &c, CI &s, float &dprev, int &sdprev, const XT &x, const float yv, const TT t,
                                        const bool lane0, Exchange &xc) {
    // inputs from the lane above (query row g*R-1); lane 0 owns query row 0 and receives the boundary instead:
    // subsequence(): C[0][j] = d + 0 (free start); std_dtw(): C[0][0] = d, then C[0][j] = d + C[0][j-1] (boundary +inf)
    float up = xc.shift(static_cast<float>(c[R - 1]));
    int sup = 0;
    if (TRACK) sup = xc.shift(static_cast<int>(s[R - 1]));
    if (STD) {
        // from column 1 on, row 0 only continues from its left neighbour.  Cost-only fill: t is wave-uniform, the boundary word is
        // rewritten once (a scalar test per step).  With tracking t is per lane (pass 2: every read has its own step), where the
        // same thing as a branch in the step cost the 16-row pass 2 over a hundred scratch reloads per four steps (r03: 19 GB of
        // FETCH_SIZE per --dtw-std launch): a select on the value instead, the word keeps its 0.
        if (TRACK)
            up = (lane0 && t > 0) ? INFINITY : up;
        else if (T0 && t == 0)
            xc.template set_boundary<TRACK>(lane0, INFINITY);
    }
    float diag = dprev;
    int sdiag = sdprev;
    dprev = up;
    sdprev = sup;
    if (STD && T0) dprev = (t == 0) ? INFINITY : dprev;  // there is no column -1
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float left = c[r];
        const int sleft = s[r];
        float m;
        if (r == 0) {
            // `up` and `diag` come out of LDS here; costs are non-negative floats, whose order is the order of their
            // bit patterns, so an unsigned min3 gives the same value without the sNaN-quieting ops fminf would add
            const unsigned mu = min(min(__float_as_uint(up), __float_as_uint(diag)), __float_as_uint(left));
            m = __uint_as_float(mu);
        } else {
            m = fminf(fminf(up, diag), left);  // v_min3_f32; no NaNs on this path
        }
        const float cn = fabsf(x[r] - yv) + m;
        int sn = 0;
        if (TRACK) {
            // traceback order of path(): diagonal first, then left, then up (src/cdtw.c:134-146)
            sn = (diag == m) ? sdiag : ((left == m) ? sleft : sup);
            if (r == 0) sn = lane0 ? static_cast<int>(t) : sn;  // query row 0: the path starts in this column
        }
        diag = left;
        sdiag = sleft;
        up = cn;
        sup = sn;
        c[r] = cn;
        s[r] = sn;
    }
}

template <int R>
__device__ __forceinline__ void load_query_rows(float (&x)[R], const DpArgs &a, int read, int qlen, int g, int g0 = 0) {
    const float *q = a.queries + a.q_off[read >= 0 ? read : 0];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = (g - g0) * R + r;  // (qlen: this read's own length; rows in front of lane g0 and past the end are zeros)
        const int src = a.rev_query ? (qlen - 1 - i) : i;
        x[r] = (read >= 0 && i >= 0 && i < qlen) ? q[src] : 0.0f;
    }
}

// The lane's query rows in LDS instead of registers: plane r of the wave's block, 64 lanes wide.  Pass 2 of the 32-row shapes inside
// the fill launch runs on that launch's 128-VGPR budget; costs, start columns and query rows (3 x 32 registers) do not fit it, and
// spilled to scratch they were reloaded in every step (85 scratch loads per four steps).  The rows are read once per cell, so they
// are the ones to go: one ds_read_b32 with an immediate offset per cell.  volatile: the loads stay where they are (hoisted out of
// the step loop they would be registers again).
struct LdsRows {
    typedef __attribute__((address_space(3))) volatile float lds_vf;
    float *p;  // this lane's column of the wave's [R][64] block
    __device__ __forceinline__ float operator[](const int r) const { return *((lds_vf *)(p + r * 64)); }
    template <int R>
    __device__ __forceinline__ void load(const DpArgs &a, int read, int qlen, int g, int g0) {
        const float *q = a.queries + a.q_off[read >= 0 ? read : 0];
        for (int r = 0; r < R; ++r) {
            const int i = (g - g0) * R + r;
            const int src = a.rev_query ? (qlen - 1 - i) : i;
            *((lds_vf *)(p + r * 64)) = (read >= 0 && i >= 0 && i < qlen) ? q[src] : 0.0f;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// Pass 1: fill (costs only).  One wave-task = (quad, chunk of jobs).
// ---------------------------------------------------------------------------------------------------------
// Time origin.  Lane lq (owner of the last query row) meets reference column 0 at step t = lq.  Steps are issued
// in blocks of four (one 16-byte load of reference levels per block), so the sweep starts at
// t_begin = lq - roundup4(lq) in (-4, 0]: the first roundup4(lq) steps are a prologue with no last-row work, and
// from then on block b covers last-row columns 4b..4b+3 with no per-step range checks.  Columns < 0 read the
// +inf padding in front of every reference array, which keeps those cells at +inf.
__device__ __forceinline__ int sweep_begin(int lq) { return lq - ((lq + kStepsPerLoad - 1) & ~(kStepsPerLoad - 1)); }

// Checkpoint record: R cost planes + dprev plane + one plane holding the number of steps elapsed since t_begin
// when the snapshot was taken (blocks are not aligned to T, so the k-th snapshot sits at the first block boundary
// with e >= k*T; the trace kernel resumes from exactly that step).
template <int R>
__device__ __forceinline__ constexpr int ck_planes() { return R + 2; }

// Rolling checkpoints in LDS (the LCK kernels).  Pass 2 only ever restores ONE snapshot per read -- the one in front of the
// window that finally wins -- yet the plain scheme spills every snapshot of every quad to HBM (13 GB per 100 000-read
// launch).  Here a wave keeps its last two snapshots (every 512 steps) in LDS and copies one to HBM only when a window has
// just become a read's best so far (a handful of times per read): the snapshot pass 2 would pick for that window, i.e. the
// last one taken at least trace_margin + 3 steps before the window's first cell.  The host caps the margin at
// 512 - qlen - 3, so that snapshot is always one of the two on hand (see save()).  A save is skipped when another task of
// the same read has already seen a strictly better score (g_best, atomic min over float bits): that window cannot win.
// What the record of a (quad, chunk) holds at the end is the snapshot for the chunk's best window whenever that window can
// be the read's winner.  If the path turns out to start before the snapshot, pass 2 backs off to the sparse HBM
// checkpoints (every coarse_every-th snapshot is also stored there, as before) and finally to the start of the strand.
constexpr int kLdsCkShift = 9;   // the SHORTEST interval between two LDS snapshots (queries up to 256 events); DpArgs::lck_shift is the batch's: the
                                 // smallest of 512 / 1024 / 2048 steps that holds a window + the head start of pass 2 (see LdsCkpt::save)
constexpr int kLdsCkPlanes = 17;  // R <= 16 costs + dprev
struct LdsCkpt {
    float *buf;        // this lane's column of the wave's two buffers: buf[(j & 1) * kLdsCkPlanes * 64 + plane * 64]
    int count;         // snapshots taken in the current job (1-based index of the last one)
    int e0, e1;        // step index of the snapshot held by buffer 0 / 1
    float *rec;        // this lane's column of the task's HBM record
    int32_t *rec_e;    // [4] per slot
    unsigned *g_best;  // this lane's read (valid where owner)
    __device__ __forceinline__ void begin_job() { count = 0; }
    template <int R, typename CV>
    __device__ __forceinline__ void snapshot(const CV &cv, float dprev, int e) {
        count += 1;
        float *b = buf + (count & 1) * (kLdsCkPlanes * 64);
#pragma unroll
        for (int r = 0; r < R; ++r) b[r * 64] = cv[r];
        b[R * 64] = dprev;
        if (count & 1)
            e1 = e;
        else
            e0 = e;
    }
    // at the end of a window that began at step e_ws (steps since t_begin; per lane: the value of the lane's read) and became the
    // best of some read(s) of the quad: improved = ballot of the lanes owning the last query row of those reads.  All 64 lanes
    // are active here.
    template <int R, int L, bool WT = false>  // WT: write-through stores (the fused launch reads the record in the same launch)
    __device__ __forceinline__ void save(unsigned long long improved, float wmin, int e_ws, int margin, int lq, int job, int shift) {
        const int lane = threadIdx.x & 63;
        const int owner_lane = (lane & ~(L - 1)) + lq;
        const bool mine = (improved >> owner_lane) & 1;
        const unsigned wb = __float_as_uint(__shfl(wmin, owner_lane));  // costs are non-negative: bit order = value order
        unsigned old = 0;
        if (mine && lane == owner_lane) old = atomicMin(g_best, wb);
        old = __shfl(old, owner_lane);
        if (!(mine && wb <= old)) return;
        // the snapshot pass 2 would choose: index kk = floor((e_ws - margin - 3) / S), S = 1 << shift, taken at the first block
        // boundary at or after kk * S.  margin <= S - wl - 3, hence kk >= count - 1: still in its buffer.
        const int from = e_ws - margin - 3;
        const int kk = from > 0 ? (from >> shift) : 0;
        if (kk > 0) {
            const float *b = buf + (kk & 1) * (kLdsCkPlanes * 64);
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                if (WT)
                    __hip_atomic_store(rec + r * 64, b[r * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    rec[r * 64] = b[r * 64];
            }
        }
        if (lane == owner_lane) {
            const int ev = kk > 0 ? ((kk & 1) ? e1 : e0) : -1;
            if (WT)
                __hip_atomic_store(rec_e + lane / L, ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                rec_e[lane / L] = ev;
        }
        (void)job;
    }
};

// One (contig,strand) sweep of a quad.  RQ >= 0: the register holding the last query row is a compile-time
// constant (hot specialisation); RQ < 0: it is the wave-uniform value rq (indexed v_mov).
template <int R, bool STD, int RQ, bool LCK = false, int L = 16, bool WT = false>
__device__ __forceinline__ void sweep_job(const DpArgs &a, const float *yp, const int rlen, const int qlen, const int lq, const int rq,
                                          const int t_begin, const float (&x)[R], const bool lane0, Exchange &xc, Top2 &top,
                                          const int job, float *ckp, const int T, IssuePriority &pr, const MixedQuad &mq,
                                          LdsCkpt *lck = nullptr, const bool owner = false) {
    constexpr bool TRACK = false;  // start columns are pass 2's business (dp_step<.., TRACK = true> in trace_core)
    typename Vec<float, R>::type cv;
    typename Vec<int, R>::type sv;  // unused (cost-only), kept for dp_step's signature
#pragma unroll
    for (int r = 0; r < R; ++r) {
        cv[r] = INFINITY;
        sv[r] = 0;
    }
    float dprev = INFINITY;
    int sdprev = 0;
    xc.template set_boundary<TRACK>(lane0, 0.0f);

    int e = 0;               // steps executed so far; the next step is t = t_begin + e
    int ck_next = T ? T : 0x7fffffff;
    const int ck_last = T ? ((rlen > 4 ? rlen - 4 : 0) >> a.ck_shift) << a.ck_shift : 0;  // last k*T that is stored
    // LCK with std_dtw: no LDS snapshots at all (SNAP false) -- the single candidate of a job is its LAST cell and its path may
    // begin anywhere in the strand, so the state a few hundred steps in front of it is no use; pass 2 starts from the sparse HBM
    // store (interval T, tens of thousands of steps) or from the start of the strand, which for a transcriptome is the same
    constexpr bool SNAP = LCK && !STD;
    if (SNAP) {
        lck->begin_job();
        ck_next = 1 << a.lck_shift;
    }
    auto maybe_checkpoint = [&]() {  // at a block boundary: snapshot the state BEFORE step t_begin + e
        if (SNAP) {
            if (e >= ck_next) {
                lck->template snapshot<R>(cv, dprev, e);
                ck_next += 1 << a.lck_shift;
                // every coarse_every-th one also goes to the sparse HBM store (what pass 2 backs off to), same format as below;
                // write-through in the fused launch, whose pass 2 reads it in the same launch, possibly from another XCD (found by
                // the fuzz campaign: a stale record there sent pass 2 off with a garbage step index)
                if (T && (lck->count & (a.coarse_every - 1)) == 0 && (lck->count << a.lck_shift) <= ck_last) {
                    if (WT) {
#pragma unroll
                        for (int r = 0; r < R; ++r) __hip_atomic_store(ckp + r * 64, static_cast<float>(cv[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(ckp + R * 64, dprev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(ckp + (R + 1) * 64, __int_as_float(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) ckp[r * 64] = cv[r];
                        ckp[R * 64] = dprev;
                        ckp[(R + 1) * 64] = __int_as_float(e);
                    }
                    ckp += ck_planes<R>() * 64;
                }
            }
        } else if (T) {
            if (e >= ck_next && ck_next <= ck_last) {
                if (WT) {  // read by the pass-2 waves of the same launch
                    store_record_wt<R>(ckp, cv, dprev, e);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) ckp[r * 64] = cv[r];
                    ckp[R * 64] = dprev;
                    ckp[(R + 1) * 64] = __int_as_float(e);
                }
                ckp += ck_planes<R>() * 64;
                ck_next += T;
            }
        }
    };

    // one step; t = 0 (std_dtw's special column of row 0) can only fall into the first four steps of the sweep (t_begin is in
    // (-4, 0]): every later step of the cost-only std_dtw fill is compiled without the test (dp_step: T0).  A macro, not a lambda:
    // wrapped in a lambda the SUBSEQUENCE fill of the fused launch came out with 564 bytes of scratch and accesses to it in the
    // kernel body (3 GB fetched + 2.6 GB written per headline launch, 27 + 15 GB per sequin launch; caught by
    // tests/test_publish_isa.py and the PMC pass) although its code is textually the plain call.
#define SFA_DP_STEP(YV, U)                                                                                                   \
    do {                                                                                                                     \
        if constexpr (STD) {                                                                                                 \
            if (e >= kStepsPerLoad)                                                                                          \
                dp_step<R, TRACK, STD, int, false>(cv, sv, dprev, sdprev, x, (YV), t_begin + e + (U), lane0, xc);            \
            else                                                                                                             \
                dp_step<R, TRACK, STD, int, true>(cv, sv, dprev, sdprev, x, (YV), t_begin + e + (U), lane0, xc);             \
        } else {                                                                                                             \
            dp_step<R, TRACK, STD, int>(cv, sv, dprev, sdprev, x, (YV), t_begin + e + (U), lane0, xc);                       \
        }                                                                                                                    \
    } while (0)
    float4u ycur = *reinterpret_cast<const float4u *>(yp);
    // ---- prologue: the last query row has not reached column 0 yet (e_main = roundup4(lq) steps) ----
    const int e_main = lq - t_begin;
    for (; e < e_main; e += kStepsPerLoad) {
        const float4u ynext = *reinterpret_cast<const float4u *>(yp + e + kStepsPerLoad);
        maybe_checkpoint();
#pragma unroll
        for (int u = 0; u < kStepsPerLoad; ++u) SFA_DP_STEP(ycur.v[u], u);
        ycur = ynext;
    }
    // ---- main: one last-row cell per step, consumed window by window (src/sigfish.c:891-901).  Blocks of four
    // steps start wherever the previous window ended (the 16-byte reference loads need no alignment), so the
    // steady-state block carries no window-end test at all. ----
    int jqv = 0;  // last-row column of the next step
    // Windows.  std_dtw has one candidate per job and keeps one query length per wave.  The subsequence fill (MIX) carries reads
    // of different lengths (MixedQuad): every read has windows of its OWN length, so the sweep is cut wherever ANY read's window
    // ends (wave-uniform: the slots' window ends live in scalar registers), and at such a point the reads whose window ends there
    // offer their candidate and start a new one.  With equal lengths that is one cut per window.
    constexpr bool MIX = !STD;
    constexpr bool CELL = CellFromFill<R, STD>::value;
    constexpr int NS = 64 / L;  // reads per wave
    const int my_slot = (threadIdx.x & 63) / L;
    int q_s[NS], ws_s[NS], we_s[NS];  // per slot: query length, first column and end of its current window
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        q_s[sl] = MIX ? __builtin_amdgcn_readlane(mq.myq, sl * L) : qlen;
        ws_s[sl] = 0;
        we_s[sl] = min(q_s[sl], rlen);
    }
    float wmin = INFINITY;
    int wpos = CELL ? 0 : -1;
    for (int col = 0; col < rlen;) {
        pr.at_window(col);
        int nxt = we_s[0];
#pragma unroll
        for (int sl = 1; sl < NS; ++sl) nxt = min(nxt, we_s[sl]);
        const int wl = STD ? rlen : nxt - col;  // std_dtw has a single candidate: one "window"
        const int nb = wl >> 2, rm = wl & 3;
        // 16-row shapes: only the window MINIMUM is kept (one v_min per step); which column attains it first is settled in pass 2
        // for the single window that wins.  32-row shapes (CELL) also keep the column of the first strict minimum
        // (src/sigfish.c:892-899).
        auto track = [&]() {
            const float cl = (RQ >= 0) ? static_cast<float>(cv[RQ >= 0 ? RQ : 0]) : static_cast<float>(cv[rq]);
            if (!CELL) {
                wmin = fminf(wmin, cl);
            } else {
                const bool lt = cl < wmin;
                wmin = lt ? cl : wmin;
                wpos = lt ? jqv : wpos;
                jqv += 1;
            }
        };
        auto block = [&](const float4u &yv) {  // four steps on the levels in yv
            maybe_checkpoint();
#pragma unroll
            for (int u = 0; u < kStepsPerLoad; ++u) {
                SFA_DP_STEP(yv.v[u], u);
                if (!STD) track();
            }
            e += kStepsPerLoad;
        };
        // two blocks per iteration so that the prefetched levels alternate between two register sets (no copies)
        int b = 0;
        for (; R < 32 && b + 1 < nb; b += 2) {  // (at R = 32 the paired body costs 66 more VGPRs than the 4 copies it saves)
            const float4u yb = *reinterpret_cast<const float4u *>(yp + e + kStepsPerLoad);
            block(ycur);
            ycur = *reinterpret_cast<const float4u *>(yp + e + kStepsPerLoad);
            block(yb);
        }
        for (; b < nb; ++b) {
            const float4u yb = *reinterpret_cast<const float4u *>(yp + e + kStepsPerLoad);
            block(ycur);
            ycur = yb;
        }
        if (rm) {  // ragged end of the window: 1..3 steps, then the next window's loads start right behind them
            const float4u ynext = *reinterpret_cast<const float4u *>(yp + e + rm);
            maybe_checkpoint();
#pragma unroll
            for (int u = 0; u < kStepsPerLoad - 1; ++u) {
                if (u < rm) {
                    SFA_DP_STEP(ycur.v[u], u);
                    if (!STD) track();
                }
            }
            e += rm;
            ycur = ynext;
        }
        if (MIX) {
            // the reads whose window ends at column nxt (a lane mask from the scalar window ends), their window's first column
            unsigned long long endmask = 0;
            int wsl = ws_s[0];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const unsigned long long lanes_of_slot = (L == 64) ? ~0ull : (((1ull << (L & 63)) - 1) << ((sl * L) & 63));
                if (we_s[sl] == nxt) endmask |= lanes_of_slot;
                if (sl > 0) wsl = (my_slot == sl) ? ws_s[sl] : wsl;
            }
            const bool ending = (endmask >> (threadIdx.x & 63)) & 1;
            // (cost-only, 16-row shapes: the window is identified by its first column)
            const bool became_best = top.offer_if(ending, wmin, CELL ? wpos : wsl, job);
            if (LCK) {
                const unsigned long long improved = __ballot(became_best && owner);
                if (improved) lck->template save<R, L, WT>(improved, wmin, wsl + e_main, a.trace_margin, lq, job, a.lck_shift);
            }
            wmin = ending ? INFINITY : wmin;
            if (CELL) wpos = ending ? nxt : wpos;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                if (we_s[sl] == nxt) {
                    ws_s[sl] = nxt;
                    we_s[sl] = min(nxt + q_s[sl], rlen);
                }
            }
        } else {  // std_dtw: the single candidate C[n-1][m-1]
            const float cl = (RQ >= 0) ? static_cast<float>(cv[RQ >= 0 ? RQ : 0]) : static_cast<float>(cv[rq]);
            top.offer(cl, rlen - 1, job);
        }
        col += wl;
    }
    pr.remaining -= rlen;
#undef SFA_DP_STEP
}

// Dispatch on the register of the last query row: compile-time constant for the cost-only subsequence fill (worth
// 5-20 % on the small-batch shapes, whose steps are short).
template <int R, bool STD, bool LCK = false, int L = 16, bool WT = false, int I = 0>
__device__ __forceinline__ void sweep_dispatch(const DpArgs &a, const float *yp, int rlen, int qlen, int lq, int rq, int t_begin,
                                               const float (&x)[R], bool lane0, Exchange &xc, Top2 &top, int job, float *ckp, int T,
                                               IssuePriority &pr, const MixedQuad &mq, LdsCkpt *lck = nullptr, bool owner = false) {
    if constexpr (STD || R > 16) {  // R = 32 keeps the indexed read: 32 more loop bodies are not worth the build time
        sweep_job<R, STD, -1, LCK, L, WT>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, pr, mq, lck, owner);
    } else {
        if (rq == I) {
            sweep_job<R, STD, I, LCK, L, WT>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, pr, mq, lck, owner);
        } else if constexpr (I + 1 < R) {
            sweep_dispatch<R, STD, LCK, L, WT, I + 1>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, pr, mq, lck, owner);
        }
    }
}

template <int R, int L, bool STD, bool LCK = false, bool FUSED = false>
__device__ __forceinline__ void fill_body(const DpArgs &a, const ClassDesc cd, const int task_local, float *lds_f, int *lds_i,
                                          float *lds_ck = nullptr) {
    const int chunk = task_local / cd.n_quads;  // chunk-major: neighbouring waves stream the same reference
    const int quad_local = task_local - chunk * cd.n_quads;
    const int quad = cd.quad_base + quad_local;
    const int lane = threadIdx.x & 63;
    const int g = lane & (L - 1);
    const int slot = lane / L;

    const int qlen = __builtin_amdgcn_readfirstlane(a.quad_qlen[quad]);  // the LONGEST read of the wave (all of them but for MixedQuad)
    const int read = a.order[quad * 4 + slot];
    const int lq = (qlen - 1) / R;  // lane / register holding the last query row (wave-uniform)
    const int rq = (qlen - 1) - lq * R;
    const int t_begin = sweep_begin(lq);
    MixedQuad mq;
    mq.template init<R>(a, read, qlen);
    const bool lane0 = (g == mq.g0);
#ifdef SFA_TASK_TIMES
    const int dbg_task = cd.task_base + task_local;
    if (a.task_times && lane == 0) a.task_times[3 * static_cast<int64_t>(dbg_task)] = wall_clock64();
#endif

    float x[R];
    load_query_rows<R>(x, a, read, mq.myq, g, mq.g0);
    Exchange xc;
    xc.init(lds_f, lds_i, threadIdx.x >> 6, slot, g, L, mq.g0);

    Top2 top;
    top.init();

    const int T = a.ck_shift ? (1 << a.ck_shift) : 0;
    const int64_t ck_total = T ? a.job_ck_off[a.chunk_begin[a.n_chunks]] : 0;

    const int jb = a.chunk_begin[chunk], je = a.chunk_begin[chunk + 1];
    IssuePriority pr;
    {
        int cols = 0;
        const bool on = a.prio_unit > 0;
        if (on)
            for (int job = jb; job < je; ++job) cols += a.job_len[job];
        // fused launch: the ticket counter already says how many tasks have begun
        pr.start(on ? a.prio_unit : 0, cols, FUSED ? a.ticket : a.started, static_cast<unsigned>(a.n_tasks), !FUSED);
    }
    LdsCkpt lck;
    if (LCK && !STD) {
        const int64_t task = static_cast<int64_t>(quad) * a.n_chunks + chunk;
        lck.buf = lds_ck + (threadIdx.x >> 6) * (2 * kLdsCkPlanes * 64) + lane;
        lck.rec = a.best_rec + task * a.best_planes * 64 + lane;
        lck.rec_e = a.best_e + task * 4;
        lck.g_best = a.g_best + (read >= 0 ? read : 0);
        lck.e0 = lck.e1 = 0;
        if (g == lq && read >= 0) {  // nothing saved yet: pass 2 starts the strand from scratch
            if (FUSED)
                __hip_atomic_store(lck.rec_e + slot, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                lck.rec_e[slot] = -1;
        }
    }
    for (int job = jb; job < je; ++job) {
        const int rlen = a.job_len[job];
        const float *yp = a.ref + a.job_off[job] - g + t_begin;  // this lane's column at step t is t-g
        float *ckp = nullptr;
        if (T) ckp = a.ck + cd.ck_base + (static_cast<int64_t>(quad_local) * ck_total + a.job_ck_off[job]) * (ck_planes<R>() * 64) + lane;
        sweep_dispatch<R, STD, LCK, L, FUSED>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, pr, mq, &lck, g == lq && read >= 0);
    }

    if (g == lq && read >= 0) {
        const int64_t o = (static_cast<int64_t>(quad) * a.n_chunks + chunk) * 4 + slot;
        if (FUSED) {  // (write-through, see below)
            __hip_atomic_store(a.p_best + o, top.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.p_second + o, top.second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.p_end + o, top.end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.p_job + o, top.job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            a.p_best[o] = top.best;
            a.p_second[o] = top.second;
            a.p_end[o] = top.end;
            a.p_job[o] = top.job;
        }
    }
    if (FUSED) {
        // Partial results, best-window records and sparse checkpoints of this task are complete and were all stored
        // write-through; wait for them, then count the quad's task in (see drain_stores()).  The textbook agent-scope release
        // fence would write back the whole L2 of the XCD at the end of every one of 50 000 tasks: measured as 2.07 GB of
        // WRITE_SIZE per launch instead of 0.3 (profiles/r02_v2 vs r02_v3).
        drain_stores();
        if (lane == 0 && quad != a.debug_drop_quad) __hip_atomic_fetch_add(a.quad_done + quad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        set_issue_priority(0);
    }
#ifdef SFA_TASK_TIMES
    if (a.task_times && lane == 0) {
        a.task_times[3 * static_cast<int64_t>(dbg_task) + 1] = wall_clock64();
        a.task_times[3 * static_cast<int64_t>(dbg_task) + 2] = simd_position();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Column segments: the latency of a small batch is one wave walking a whole strand (30 k dependent steps).  A segment
// wave starts `warm_windows` windows before its own first window FROM THE JOB'S INITIAL STATE (all costs +inf, free start
// in row 0) -- a guess: the true state there depends on everything before.  But accumulated costs of paths that began
// more than a few query lengths back never survive the min, so after the warm-up the anti-diagonal state is normally
// the true one, and that is CHECKED: the wave stores its state where its own windows begin, the wave of the previous
// segment stores its state where it ends (the same step of the same sweep), and sdtw_verify_kernel compares the two
// bit for bit.  Equal state => everything the segment computed from there on is what the single sweep computes (every
// cell is a pure function of its neighbours).  Any mismatch flags the quad and the host re-runs the batch unsegmented.
// Windows, blocks of four steps and checkpoints fall on the same steps as in the single sweep because segments start
// on window boundaries; checkpoints and candidates of the warm-up are simply not stored.
__device__ __forceinline__ int64_t verify_slot(const DpArgs &a, int quad, int job, int seg, int which) {
    return ((((static_cast<int64_t>(quad) * a.n_jobs + job) * a.n_seg + seg) * 2 + which) * a.verify_planes) * 64;
}

template <int R, int RQ>
__device__ __forceinline__ void sweep_segment(const DpArgs &a, const float *yp, const int rlen, const int qlen, const int lq, const int rq,
                                              const int t_begin, const float (&x)[R], const bool lane0, Exchange &xc, Top2 &top,
                                              const int job, float *ckp, const int T, const int col0, const int col_real, const int col_end,
                                              float *vin, float *vout) {
    typename Vec<float, R>::type cv;
    typename Vec<int, R>::type sv;  // unused (cost-only), kept for dp_step's signature
#pragma unroll
    for (int r = 0; r < R; ++r) {
        cv[r] = INFINITY;
        sv[r] = 0;
    }
    float dprev = INFINITY;
    int sdprev = 0;
    xc.template set_boundary<false>(lane0, 0.0f);

    const int e_pro = lq - t_begin;  // prologue length of the single sweep: the last row is at column e - e_pro
    int e = col0;                    // global step index: the sweep is entered where the single sweep is after col0 columns
    const int e_real = col_real + e_pro;
    int ck_next = T ? T * (e / T + 1) : 0x7fffffff;
    if (T) ckp += static_cast<int64_t>(e / T) * (ck_planes<R>() * 64);
    const int ck_last = T ? ((rlen > 4 ? rlen - 4 : 0) >> a.ck_shift) << a.ck_shift : 0;
    auto maybe_checkpoint = [&]() {
        if (T) {
            if (e >= ck_next && ck_next <= ck_last) {
                if (e >= e_real) {  // (before that the state is the guess: the previous segment owns these records)
#pragma unroll
                    for (int r = 0; r < R; ++r) ckp[r * 64] = cv[r];
                    ckp[R * 64] = dprev;
                    ckp[(R + 1) * 64] = __int_as_float(e);
                }
                ckp += ck_planes<R>() * 64;
                ck_next += T;
            }
        }
    };
    auto snapshot = [&](float *v) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r * 64] = cv[r];
        v[R * 64] = dprev;
    };

    float4u ycur = *reinterpret_cast<const float4u *>(yp + e);
    for (; e < col0 + e_pro; e += kStepsPerLoad) {
        const float4u ynext = *reinterpret_cast<const float4u *>(yp + e + kStepsPerLoad);
        maybe_checkpoint();
#pragma unroll
        for (int u = 0; u < kStepsPerLoad; ++u) dp_step<R, false, false, int>(cv, sv, dprev, sdprev, x, ycur.v[u], t_begin + e + u, lane0, xc);
        ycur = ynext;
    }
    for (int col = col0; col < col_end;) {
        if (col == col_real && vin) snapshot(vin);
        const int wl = min(qlen, rlen - col);
        const int nb = wl >> 2, rm = wl & 3;
        constexpr bool CELL = CellFromFill<R, false>::value;
        float wmin = INFINITY;
        int wpos = col, jq = col;  // (CELL) column of the window's first strict minimum / of the next last-row cell
        auto track = [&]() {
            const float cl = (RQ >= 0) ? static_cast<float>(cv[RQ >= 0 ? RQ : 0]) : static_cast<float>(cv[rq]);
            if (!CELL) {
                wmin = fminf(wmin, cl);
            } else {
                const bool lt = cl < wmin;
                wmin = lt ? cl : wmin;
                wpos = lt ? jq : wpos;
                jq += 1;
            }
        };
        auto block = [&](const float4u &yv) {
            maybe_checkpoint();
#pragma unroll
            for (int u = 0; u < kStepsPerLoad; ++u) {
                dp_step<R, false, false, int>(cv, sv, dprev, sdprev, x, yv.v[u], t_begin + e + u, lane0, xc);
                track();
            }
            e += kStepsPerLoad;
        };
        for (int b = 0; b < nb; ++b) {
            const float4u yb = *reinterpret_cast<const float4u *>(yp + e + kStepsPerLoad);
            block(ycur);
            ycur = yb;
        }
        if (rm) {
            const float4u ynext = *reinterpret_cast<const float4u *>(yp + e + rm);
            maybe_checkpoint();
#pragma unroll
            for (int u = 0; u < kStepsPerLoad - 1; ++u) {
                if (u < rm) {
                    dp_step<R, false, false, int>(cv, sv, dprev, sdprev, x, ycur.v[u], t_begin + e + u, lane0, xc);
                    track();
                }
            }
            e += rm;
            ycur = ynext;
        }
        if (col >= col_real) top.offer(wmin, CELL ? wpos : col, job);
        col += wl;
    }
    if (vout) snapshot(vout);
}

template <int R, int I = 0>
__device__ __forceinline__ void segment_dispatch(const DpArgs &a, const float *yp, int rlen, int qlen, int lq, int rq, int t_begin,
                                                 const float (&x)[R], bool lane0, Exchange &xc, Top2 &top, int job, float *ckp, int T,
                                                 int col0, int col_real, int col_end, float *vin, float *vout) {
    if constexpr (R > 16) {
        sweep_segment<R, -1>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, col0, col_real, col_end, vin, vout);
    } else {
        if (rq == I) {
            sweep_segment<R, I>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, col0, col_real, col_end, vin, vout);
        } else if constexpr (I + 1 < R) {
            segment_dispatch<R, I + 1>(a, yp, rlen, qlen, lq, rq, t_begin, x, lane0, xc, top, job, ckp, T, col0, col_real, col_end, vin, vout);
        }
    }
}

// how segment `seg` of a job of rlen columns looks for a quad of query length qlen (also used by the verify kernel)
struct SegRange {
    int col0, col_real, col_end;
    bool empty, last;
};
__device__ __forceinline__ SegRange segment_range(int rlen, int qlen, int n_seg, int warm_windows, int seg) {
    const int nwin = (rlen + qlen - 1) / qlen;
    const int nw = (nwin + n_seg - 1) / n_seg;
    const int w_real = seg * nw;
    SegRange r;
    r.empty = w_real >= nwin;
    r.last = w_real + nw >= nwin;
    r.col_real = w_real * qlen;
    r.col0 = max(0, w_real - warm_windows) * qlen;
    r.col_end = min(rlen, (w_real + nw) * qlen);
    return r;
}

// wave-task = (quad, job, segment); cost-only subsequence DTW (small batches)
template <int R, int L>
__device__ __forceinline__ void fill_body_seg(const DpArgs &a, const ClassDesc cd, const int task_local, float *lds_f, int *lds_i) {
    const int chunk = task_local / cd.n_quads;
    const int quad_local = task_local - chunk * cd.n_quads;
    const int quad = cd.quad_base + quad_local;
    const int lane = threadIdx.x & 63;
    const int g = lane & (L - 1);
    const int slot = lane / L;
    const int job = chunk / a.n_seg, seg = chunk - job * a.n_seg;
    const int qlen = __builtin_amdgcn_readfirstlane(a.quad_qlen[quad]);
    const int read = a.order[quad * 4 + slot];
    const int lq = (qlen - 1) / R;
    const int rq = (qlen - 1) - lq * R;
    const int t_begin = sweep_begin(lq);
    Top2 top;
    top.init();
    const int rlen = a.job_len[job];
    const SegRange sr = segment_range(rlen, qlen, a.n_seg, a.warm_windows, seg);
    if (!sr.empty) {
        float x[R];
        load_query_rows<R>(x, a, read, qlen, g);
        Exchange xc;
        xc.init(lds_f, lds_i, threadIdx.x >> 6, slot, g, L);
        const int T = a.ck_shift ? (1 << a.ck_shift) : 0;
        const int64_t ck_total = T ? a.job_ck_off[a.n_jobs] : 0;
        const float *yp = a.ref + a.job_off[job] - g + t_begin;
        float *ckp = nullptr;
        if (T) ckp = a.ck + cd.ck_base + (static_cast<int64_t>(quad_local) * ck_total + a.job_ck_off[job]) * (ck_planes<R>() * 64) + lane;
        float *vin = seg > 0 ? a.verify + verify_slot(a, quad, job, seg, 0) + lane : nullptr;
        float *vout = !sr.last ? a.verify + verify_slot(a, quad, job, seg, 1) + lane : nullptr;
        segment_dispatch<R>(a, yp, rlen, qlen, lq, rq, t_begin, x, g == 0, xc, top, job, ckp, T, sr.col0, sr.col_real, sr.col_end, vin, vout);
    }
    if (g == lq && read >= 0) {
        const int64_t o = (static_cast<int64_t>(quad) * a.n_chunks + chunk) * 4 + slot;
        a.p_best[o] = top.best;
        a.p_second[o] = top.second;
        a.p_end[o] = top.end;
        a.p_job[o] = top.job;
    }
}

// grid: ceil(n_tasks/4) blocks of 256 threads (4 waves, one task each).  MAXR (rows per lane) bounds the shapes
// compiled in, so a batch without long queries does not pay the long variant's register budget.
// SEG: the column-segment variant (cost-only sDTW, small batches) is its own kernel, so that the throughput kernel's code
// is not touched by it.
// pass 2 of one quad inside the fill launch (FUSED kernels; defined behind the trace code)
// (not inlined: its register needs -- 128 VGPRs and some spills -- then stay its own business instead of leaning on the
// allocation of the fill loops it shares the kernel with; and its arguments come from a copy of the kernarg block in DEVICE
// memory (`a.self`): a reference to the kernel's own parameter would force the kernel to keep a copy of it in scratch, and
// the fill loops would then read `a.xyz` from there, per lane, instead of from scalar registers -- measured as a scratch
// load in every block of four steps; passing the 600-byte block by value costs a per-LANE stack copy per call instead,
// 0.95 GB of scratch writes per 100 000-read launch)
template <int MAXR, bool STD, bool LCK>
__device__ __attribute__((noinline)) void fused_trace_dispatch(const DpArgs *pa, const int quad, float *lds_f, int *lds_i, float *lds_x);

// LCK: rolling checkpoints in LDS (LdsCkpt) -- two snapshots of 17 planes per wave, 34 KB per block, four blocks per CU.
// FUSED (with LCK, or on the 32-row fill, whose snapshots go to HBM write-through): pass 2 rides in the same launch.  Waves claim TICKETS from a counter instead of deriving their task from
// blockIdx: tickets below n_tasks are the fill tasks, in the usual order; ticket n_tasks + q is pass 2 of quad q, which waits
// until every fill task of that quad has signalled completion, merges their partial results (what sdtw_finalize_kernel
// does between the launches otherwise), recovers the start columns and writes the quad's rows.  Because a ticket is only
// claimed by a wave that is running, everything a pass-2 wave waits for is held by a resident wave: the wait cannot
// deadlock whatever order the hardware starts blocks in.  The point: when the fill's last tasks drain, SIMDs go idle one
// after the other for ~2 ms; the pass-2 tickets are claimed exactly then, so pass 2 (3 ms as its own
// launch) runs in that slack, at the lowest issue priority, and two launches + the kernel boundaries disappear.
template <int MAXR, bool STD, bool SEG = false, bool LCK = false, bool FUSED = false>
__global__ void __launch_bounds__(256, LCK ? SFA_LCK_WAVES : (MAXR <= 16 ? SFA_FILL_WAVES : (STD ? 1 : SFA_FILL32_WAVES))) sdtw_fill_kernel(const DpArgs a) {
    static_assert(!SEG || !STD, "segments: subsequence DTW");
    static_assert(!LCK || (!SEG && MAXR <= 16), "LDS checkpoints: R <= 16 (std_dtw: sparse HBM store only)");
    static_assert(!FUSED || LCK || (MAXR == 32 && !SEG && !STD), "pass 2 by ticket: on the LDS-checkpoint fill, or on the 32-row fill (snapshots in HBM)");
    // blockIdx -> task.  Blocks are dealt to the 8 XCDs round-robin, so with the identity every XCD sees every class and
    // every chunk of the job list evenly -- what this kernel wants: the reference arrays (hundreds of KB to a few MB) stay
    // resident in every XCD's L2 anyway, whereas the classes differ in speed.  Measured against XCD-contiguous mappings (LABNOTES.md,
    // fill ms): identity 75.9 / contiguous per class 76.2 / contiguous over the grid 77.7 on the headline workload, 269 / 275 /
    // 278 on RNA004 --dtw-std, 2 438 / 2 438 / 2 474 on the 1 Mb reference.  (One contiguous range per XCD over the whole grid hands
    // all the short, fast classes at the end of the task list to the last XCD, which then idles while the other seven finish.)
    int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    __shared__ float lds_f[4 * kXchWordsPerWave];
    __shared__ int lds_i[FUSED ? 4 * kXchWordsPerWave : 1];
    __shared__ float lds_ck[(LCK && !STD) ? 4 * 2 * kLdsCkPlanes * 64 : 1];
    __shared__ float lds_x[(FUSED && MAXR == 32) ? 4 * 32 * 64 : 1];  // query rows of the 32-row shapes' pass 2 (LdsRows): 32 KB per block, four blocks per CU
    if (FUSED) {  // one ticket per wave (a wave that claimed the next one itself measured level: LABNOTES.md)
        unsigned t = 0;
        if ((threadIdx.x & 63) == 0) t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        task = __builtin_amdgcn_readfirstlane(t);
        if (task >= a.n_tasks) {  // pass 2 of quad (task - n_tasks)
            const int quad = task - a.n_tasks;
            if (quad >= a.n_quads_total) return;
            fused_trace_dispatch<MAXR, STD, LCK>(a.self, quad, lds_f, lds_i, lds_x);
            return;
        }
    }
    if (task >= a.n_tasks) return;  // wave-uniform
    int ci = 0;
    while (ci + 1 < a.n_cls && task >= a.cls[ci + 1].task_base) ++ci;
    const ClassDesc cd = a.cls[ci];
    const int tl = task - cd.task_base;
#define SFA_SHAPE(RR, LL)                                                                \
    case (RR) * 256 + (LL):                                                              \
        if constexpr (MAXR >= (RR)) {                                                    \
            if constexpr (SEG)                                                           \
                fill_body_seg<RR, LL>(a, cd, tl, lds_f, lds_i); /* (quad, job, segment) */ \
            else                                                                         \
                fill_body<RR, LL, STD, LCK, FUSED>(a, cd, tl, lds_f, lds_i, lds_ck);     \
        }                                                                                \
        break;
    switch (cd.R * 256 + cd.lanes) {
        SFA_SHAPE(32, 64) SFA_SHAPE(32, 32) SFA_SHAPE(32, 16)
        SFA_SHAPE(16, 64) SFA_SHAPE(16, 32) SFA_SHAPE(16, 16)
        SFA_SHAPE(8, 64) SFA_SHAPE(8, 32) SFA_SHAPE(8, 16)
        SFA_SHAPE(4, 64) SFA_SHAPE(4, 32)
        SFA_SHAPE(4, 16)
        default:
            break;
    }
#undef SFA_SHAPE
}

// Cost and start column of register rq (wave-uniform) of this lane's rows, for the 32-row shapes: a scalar branch tree over constant
// indices, not `c[rq]`.  The indexed move wants its 32 registers in one aligned block, and where the allocator could not arrange
// that (one of the three 32-row shapes of pass 2 inside the fill launch) the whole tuple went through scratch, 672 reloads per four
// steps: 276 ms per step instead of 79.5 at q = 500.  Shapes of up to 16 rows per lane take the indexed move (76.6 -> 75.8 ms on the
// headline against the select chain a DIVERGENT index gave in rounds 2-3; the tree: 76.2).  profiles/r04_logs/ab_pass2_last_row_pick.log
template <int R, int LO = 0, int HI = R, typename CF, typename CI>
__device__ __forceinline__ void pick_row(const CF &c, const CI &s, const int rq, float &cl, int &sl) {
    if constexpr (HI - LO == 1) {
        cl = c[LO];
        sl = s[LO];
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (rq < MID)
            pick_row<R, LO, MID>(c, s, rq, cl, sl);
        else
            pick_row<R, MID, HI>(c, s, rq, cl, sl);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pass 2: start-column recovery for each read's winning candidate.  One wave = one quad; each 16-lane row
// follows its own (job, end column, checkpoint), so the step index is per lane.
// ---------------------------------------------------------------------------------------------------------
struct ResultRow;  // below

// the winner of a read as pass 2 needs it (per lane: the value of the lane's read)
struct Winner {
    int job;     // -1: nothing to trace
    int ws;      // first column of the winning window
    float best;  // the winning score
    int chunk;   // LCK: the task whose record holds the snapshot
};

template <int R, int L, bool STD, bool LCK = false, bool XLDS = false>
__device__ __forceinline__ void trace_core(const DpArgs &a, const ClassDesc cd, const int quad_local, const Winner w, float *lds_f, int *lds_i,
                                           int &res_st, int &res_end, float *lds_x = nullptr);

template <int R, int L, bool STD, bool LCK = false>
__device__ __forceinline__ void trace_body(const DpArgs &a, const ClassDesc cd, const int quad_local, int32_t *out_st, float *lds_f,
                                           int *lds_i) {
    const int quad = cd.quad_base + quad_local;
    const int lane = threadIdx.x & 63;
    const int slot = lane / L;
    const int read = a.order[quad * 4 + slot];
    Winner w;
    w.job = (read >= 0) ? a.w_job[read] : -1;
    w.ws = (read >= 0) ? a.w_end[read] : 0;
    w.best = (read >= 0) ? a.w_score[read] : 0.0f;
    w.chunk = (LCK && read >= 0) ? a.w_chunk[read] : 0;
    int res_st, res_end;
    trace_core<R, L, STD, LCK>(a, cd, quad_local, w, lds_f, lds_i, res_st, res_end);
    const int qlen = a.quad_qlen[quad];
    if ((lane & (L - 1)) == (qlen - 1) / R && read >= 0) {
        out_st[read] = res_st;
        out_st[a.n_reads_total + read] = res_end;
    }
}

template <int R, int L, bool STD, bool LCK, bool XLDS>
__device__ __forceinline__ void trace_core(const DpArgs &a, const ClassDesc cd, const int quad_local, const Winner w, float *lds_f, int *lds_i,
                                           int &res_st, int &res_end, float *lds_x) {
    const int quad = cd.quad_base + quad_local;
    const int lane = threadIdx.x & 63;
    const int g = lane & (L - 1);
    const int slot = lane / L;

    // the longest read of the wave: lane and register of everybody's last query row.  Wave-uniform, and SAID so: inside the fused
    // launch the arguments arrive through a pointer in vector registers, the compiler takes everything loaded through it for
    // divergent, and `c[rq]` with a divergent index is a chain of R compare + select pairs per step (4 v_cndmask per cell instead
    // of 2 in the shipped pass 2 of rounds 2-3); with a scalar index it is one indexed move
    const int qlen = __builtin_amdgcn_readfirstlane(a.quad_qlen[quad]);
    const int read = a.order[quad * 4 + slot];
    const int lq = (qlen - 1) / R;
    const int rq = (qlen - 1) - lq * R;
    MixedQuad mq;  // this read's own length and the lane its query row 0 sits in
    mq.template init<R>(a, read, qlen);
    const bool lane0 = (g == mq.g0);

    float x[XLDS ? 1 : R];
    LdsRows xl;  // XLDS: the rows live in LDS (lds_x: [4 waves][R][64])
    xl.p = XLDS ? lds_x + (threadIdx.x >> 6) * (R * 64) + lane : nullptr;
    if constexpr (XLDS)
        xl.template load<R>(a, read, mq.myq, g, mq.g0);
    else
        load_query_rows<R>(x, a, read, mq.myq, g, mq.g0);
    Exchange xc;
    xc.init(lds_f, lds_i, threadIdx.x >> 6, slot, g, L, mq.g0);

    int job = (read >= 0) ? w.job : -1;
    constexpr bool CELL = CellFromFill<R, STD>::value && !LCK;
    const int ws = (read >= 0) ? w.ws : 0;  // first column of the winning window; CELL: the column of the winning cell itself
    const float best = (read >= 0) ? w.best : 0.0f;
    bool done = !(read >= 0 && job >= 0 && ws >= 0);
    job = done ? 0 : job;
    const int rlen = a.job_len[job];
    const int wl = STD ? 1 : min(mq.myq, rlen - ws);  // the read's own window length; std_dtw: the "window" is the last column alone
    const int t_begin = sweep_begin(lq);             // same time origin as the fill
    const float *ybase = a.ref + a.job_off[job] - g;
    const int t_first = ws + lq;            // step at which lane lq evaluates the first cell of the window
    const int t_last = CELL ? t_first : ws + wl - 1 + lq;  // ... and the last one

    const int T = a.ck_shift ? (1 << a.ck_shift) : 0;
    const int64_t ck_total = a.ck_shift ? a.job_ck_off[a.chunk_begin[a.n_chunks]] : 0;
    const int nck = T ? (rlen > 4 ? rlen - 4 : 0) >> a.ck_shift : 0;  // checkpoints stored for this job
    int k = 0;
    if (T) {
        // snapshot k sits at most 3 steps after k*T, and it must not lie behind the window
        const int from = t_first - a.trace_margin - t_begin - 3;  // steps elapsed since t_begin
        k = from > 0 ? (from >> a.ck_shift) : 0;
        k = k < nck ? k : nck;
    }
    int back = 1;
    res_st = -1;
    res_end = -1;
    const unsigned long long owner = __ballot(g == lq);  // the lanes that own a last query row
    // LCK: the first attempt restores the snapshot the fill saved for the winning window (the record of the read's winning
    // chunk); only if the path began before it do the sparse HBM checkpoints (k, as computed above with their interval) come in
    bool use_rec = false;
    const float *recp = nullptr;
    int rec_e = -1;
    if (LCK && !STD && !done) {  // (std_dtw keeps no records: k, into the sparse store, stands)
        const int64_t task = static_cast<int64_t>(quad) * a.n_chunks + w.chunk;
        rec_e = a.best_e[task * 4 + slot];
        recp = a.best_rec + task * a.best_planes * 64 + lane;
        use_rec = rec_e >= 0;
        if (!use_rec) k = 0;  // the fill chose "from scratch" for this window (it lies within the first 512 + margin steps)
    }

    for (int attempt = 0; attempt < 48; ++attempt) {  // bounded: k reaches 0 after <= 32 halvings
        int tb = t_begin;  // first step to execute
        typename Vec<float, R>::type c;
        typename Vec<int, R>::type s;
        float dprev;
        int sdprev;
        if (LCK && use_rec) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c[r] = recp[r * 64];
                s[r] = -1;
            }
            dprev = recp[R * 64];
            sdprev = -1;
            tb = t_begin + rec_e;
        } else if (k > 0) {  // restore the exact anti-diagonal state; provenance of these cells is unknown (-1)
            const float *ckp = a.ck + cd.ck_base +
                               (static_cast<int64_t>(quad_local) * ck_total + a.job_ck_off[job] + (k - 1)) * (ck_planes<R>() * 64) + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c[r] = ckp[r * 64];
                s[r] = -1;
            }
            dprev = ckp[R * 64];
            sdprev = -1;
            tb = t_begin + __float_as_int(ckp[(R + 1) * 64]);  // the step this snapshot was taken before
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c[r] = INFINITY;
                s[r] = 0;
            }
            dprev = INFINITY;
            sdprev = 0;
        }
        // std_dtw(): a row that restarts at or before column 0 begins with the free boundary, later ones with +inf
        xc.template set_boundary<true>(lane0, (STD && tb > 0) ? INFINITY : 0.0f);
        // (a step index behind the window's first cell, or in front of the sweep's origin, cannot be this window's snapshot: the
        // lane sits the attempt out -- from a harmless position -- and backs off)
        const bool bad_rec = !done && (tb > t_first || tb < t_begin);
        if (bad_rec) tb = t_begin;
        const int len = (done || bad_rec) ? 0 : (t_last - tb + 1);
        int maxlen = __builtin_amdgcn_readlane(len, 0);
        if (L <= 32) maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 32));
        if (L <= 16) {
            maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 16));
            maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 48));
        }

        // first cell of the window whose cost equals the winning score (= the first strict minimum the reference's
        // scan selects, src/sigfish.c:892-899), and the start column carried into it
        int cap_end = done ? 0 : -1, cap_st = -1;
        const int tlim = rlen + 64;  // keep the loads of rows that are already done inside the padded array
        for (int tau0 = 0; tau0 < maxlen; tau0 += kStepsPerLoad) {
            const int tbl = min(tb + tau0, tlim);
            const float4u yv = *reinterpret_cast<const float4u *>(ybase + tbl);
#pragma unroll
            for (int u = 0; u < kStepsPerLoad; ++u) {
                const int t = tb + tau0 + u;
                if constexpr (XLDS)
                    dp_step<R, true, STD, int>(c, s, dprev, sdprev, xl, yv.v[u], t - mq.g0, lane0, xc);
                else
                    dp_step<R, true, STD, int>(c, s, dprev, sdprev, x, yv.v[u], t - mq.g0, lane0, xc);  // (the column of the lane holding row 0)
                float cl;
                int sl;
                if constexpr (R >= 32) {
                    pick_row<R>(c, s, rq, cl, sl);
                } else {  // one indexed move each (s_set_gpr_idx): rq is a scalar
                    cl = c[rq];
                    sl = s[rq];
                }
                const bool hit = (cap_end < 0) && !bad_rec && (t >= t_first) && (t <= t_last) && (cl == best);
                cap_end = hit ? (t - lq) : cap_end;
                cap_st = hit ? sl : cap_st;
            }
            if ((__ballot(cap_end >= 0) & owner) == owner) break;  // every read of the quad has its cell
        }
        const int src = (lane & ~(L - 1)) + lq;  // the lane that owns the last query row of this read
        const int q_end = __shfl(cap_end, src), q_st = __shfl(cap_st, src);
        if (!done) {
            if ((q_end >= 0 && q_st >= 0) || (k == 0 && !(LCK && use_rec))) {
                res_end = q_end;
                res_st = q_st;
                done = true;
            } else if (LCK && use_rec) {  // the path starts before the saved snapshot: on to the sparse store (k), then 0
                use_rec = false;
            } else {  // the path starts before this checkpoint: back off (1, 2, 4, ... checkpoints)
                k = max(0, k - back);
                back <<= 1;
            }
        }
        if (__all(done)) break;
    }
}

// One result row per read, POD mirror of sfa_result_t (include/sigfish_amd.h).
struct ResultRow {
    int32_t rid, pos_st, pos_end;
    float score, score2;
    int8_t strand;
    uint8_t mapq, valid, pad;
};

struct FinalizeArgs {
    const int32_t *slot_of_read;  // [n_reads] quad*4+slot, or -1 for skipped reads
    const float *p_best;
    const int32_t *p_end;
    const int32_t *p_job;
    const float *p_second;
    const int32_t *job_contig;  // [n_jobs]
    const int8_t *job_strand;   // [n_jobs] '+' / '-'
    const int32_t *ref_len;     // [num_ref]
    const int32_t *ref_st_offset;
    int32_t *w_job;  // winners for the trace kernel
    int32_t *w_end;
    float *w_score;
    int32_t *w_chunk;  // chunk (task) the winner came from: where the LCK fill left its snapshot
    const int32_t *t_st;  // second finalize: start columns [n_reads] then end columns [n_reads] from the trace kernel
    ResultRow *out;       // [n_reads]
    const uint8_t *bad;   // [n_reads] 1: a query value is NaN or +-inf (sdtw_screen_kernel) -> the read is skipped
    const int64_t *q_off;  // [n_reads+1]; with max_query: a longer read belongs to the row-strip path (sdtw_strips.hpp), which
    int32_t max_query;     // writes its row from another stream -- no row is written for it here (0: every row is written)
    int32_t n_reads, n_chunks;
    int32_t mode;  // 1: after fill -> winners + scores; 2: after trace -> positions;
                   // 3: fused launch -> rows of the reads that are in no quad (skipped), the rest is written by its pass-2 waves
    // mode 2: how many reference columns the alignments of this batch span, in sixteenths of their query length (32 buckets, the
    // last one open): what the next batch's pass 2 takes as its head start instead of a whole query length (nullptr: not kept)
    unsigned *span_hist;
};
template <int MAXR, bool STD, bool LCK = false>
__global__ void __launch_bounds__(256, MAXR <= 16 ? SFA_TRACE_WAVES : 1) sdtw_trace_kernel(const DpArgs a, int32_t *out_st) {
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (task >= a.n_tasks) return;
    int ci = 0;
    while (ci + 1 < a.n_cls && task >= a.cls[ci + 1].task_base) ++ci;
    const ClassDesc cd = a.cls[ci];
    const int tl = task - cd.task_base;
    __shared__ float lds_f[4 * kXchWordsPerWave];
    __shared__ int lds_i[4 * kXchWordsPerWave];
#define SFA_SHAPE(RR, LL)                                                                    \
    case (RR) * 256 + (LL):                                                                  \
        if constexpr (MAXR >= (RR)) trace_body<RR, LL, STD, LCK>(a, cd, tl, out_st, lds_f, lds_i); \
        break;
    switch (cd.R * 256 + cd.lanes) {
        SFA_SHAPE(32, 64) SFA_SHAPE(32, 32) SFA_SHAPE(32, 16)
        SFA_SHAPE(16, 64) SFA_SHAPE(16, 32) SFA_SHAPE(16, 16)
        SFA_SHAPE(8, 64) SFA_SHAPE(8, 32) SFA_SHAPE(8, 16)
        SFA_SHAPE(4, 64) SFA_SHAPE(4, 32)
        default:
            trace_body<4, 16, STD, LCK>(a, cd, tl, out_st, lds_f, lds_i);
            break;
    }
#undef SFA_SHAPE
}

// src/sigfish.c:979-983: (int)round(500*(score2-score)/score) with x86 cvttsd2si saturation, cap 60, store u8
__device__ __forceinline__ uint8_t mapq_from_scores(float score, float score2) {
    const float v = 500.0f * (score2 - score) / score;
    const float r = roundf(v);  // == round((double)v) for float inputs
    int q;
    if (!(r >= -2147483648.0f && r < 2147483648.0f))
        q = INT32_MIN;
    else
        q = static_cast<int>(r);
    if (q > 60) q = 60;
    return static_cast<uint8_t>(q);
}

template <int R, int L, bool STD, bool LCK>
__device__ __forceinline__ void fused_trace_task(const DpArgs &a, const ClassDesc cd, const int quad_local, float *lds_f, int *lds_i, float *lds_x) {
    const int quad = cd.quad_base + quad_local;
    const int lane = threadIdx.x & 63;
    const int g = lane & (L - 1);
    const int slot = lane / L;
    // every fill task of the quad must have published its results (they hold lower tickets: running or finished)
    int seen;
    const bool arrived = bounded_wait_ge(a.quad_done + quad, a.n_chunks, a.spin_limit, &seen, true);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (on the give-up path too: every load behind the poll is behind the acquire)
    __builtin_amdgcn_wave_barrier();
    if (!arrived) {
        if (lane == 0) report_device_error(a.err, kErrQuadWait, quad, seen);
        return;  // wave-uniform; the host reports SFA_EKERNEL for the batch
    }
    const int qlen = __builtin_amdgcn_readfirstlane(a.quad_qlen[quad]);
    const int lq = (qlen - 1) / R;
    const int read = a.order[quad * 4 + slot];
    const bool owner = (g == lq) && read >= 0;
    // merge of the chunks' partial top-2 in processing order (sdtw_finalize_kernel, mode 1), one lane per read
    ResultRow r;
    r.rid = -1;
    r.pos_st = -1;
    r.pos_end = -1;
    r.score = INFINITY;
    r.score2 = INFINITY;
    r.strand = 0;
    r.mapq = 0;
    r.valid = 0;
    r.pad = 0;
    Winner w;
    w.job = -1;
    w.ws = -1;
    w.best = 0.0f;
    w.chunk = 0;
    if (owner && !a.bad[read]) {
        float best = INFINITY, second = INFINITY;
        int end = -1, job = -1, chunk = 0;
        for (int ch = 0; ch < a.n_chunks; ++ch) {  // a later chunk wins ties
            const int64_t o = (static_cast<int64_t>(quad) * a.n_chunks + ch) * 4 + slot;
            const float b = a.p_best[o], s2 = a.p_second[o];
            const float hi = fmaxf(best, b);
            const float lo2 = fminf(second, s2);
            const bool take = !(b > best);
            second = fminf(hi, lo2);
            if (take) {
                best = b;
                end = a.p_end[o];
                job = a.p_job[o];
                chunk = ch;
            }
        }
        r.valid = 1;
        r.score = best;
        r.score2 = second;
        if (job >= 0) {
            r.rid = a.job_contig[job];
            r.strand = a.job_strand[job];
            r.mapq = mapq_from_scores(best, second);
            w.job = job;
            w.ws = end;
            w.best = best;
            w.chunk = chunk;
        }
    }
    const int src = (lane & ~(L - 1)) + lq;  // the read's owner lane
    w.job = __shfl(w.job, src);
    w.ws = __shfl(w.ws, src);
    w.best = __shfl(w.best, src);
    w.chunk = __shfl(w.chunk, src);
    int res_st, res_end;
    trace_core<R, L, STD, LCK, (R == 32)>(a, cd, quad_local, w, lds_f, lds_i, res_st, res_end, lds_x);
    if (owner) {
        if (r.rid >= 0) {  // src/sigfish.c:971-975
            const int rl = a.ref_len[r.rid], off = a.ref_st_offset[r.rid];
            r.pos_st = ((r.strand == '+') ? res_st : rl - res_end) + off;
            r.pos_end = ((r.strand == '+') ? res_end : rl - res_st) + off;
            // how many reference columns this alignment spans, in sixteenths of its query length (sdtw_finalize_kernel, mode 2)
            if (a.span_hist && res_st >= 0 && res_end >= res_st) {
                const int64_t ql = a.q_off[read + 1] - a.q_off[read];
                if (ql > 0) {
                    const int64_t b = (static_cast<int64_t>(res_end - res_st + 1) * 16) / ql;
                    atomicAdd(a.span_hist + (b < kSpanBuckets - 1 ? static_cast<int>(b) : kSpanBuckets - 1), 1u);
                }
            }
        }
        a.out[read] = r;
    }
}

template <int MAXR, bool STD, bool LCK>
__device__ __attribute__((noinline)) void fused_trace_dispatch(const DpArgs *pa, const int quad, float *lds_f, int *lds_i, float *lds_x) {
    const DpArgs &a = *pa;
    int ci = 0;
    while (ci + 1 < a.n_cls && quad >= a.cls[ci + 1].quad_base) ++ci;
    const ClassDesc cd = a.cls[ci];
    const int ql = quad - cd.quad_base;
#define SFA_TSHAPE(RR, LL)                                                               \
    case (RR) * 256 + (LL):                                                              \
        if constexpr (MAXR >= (RR)) fused_trace_task<RR, LL, STD, LCK>(a, cd, ql, lds_f, lds_i, lds_x); \
        break;
    switch (cd.R * 256 + cd.lanes) {
        SFA_TSHAPE(32, 64) SFA_TSHAPE(32, 32) SFA_TSHAPE(32, 16)
        SFA_TSHAPE(16, 64) SFA_TSHAPE(16, 32) SFA_TSHAPE(16, 16)
        SFA_TSHAPE(8, 64) SFA_TSHAPE(8, 32) SFA_TSHAPE(8, 16)
        SFA_TSHAPE(4, 64) SFA_TSHAPE(4, 32) SFA_TSHAPE(4, 16)
        default:
            break;
    }
#undef SFA_TSHAPE
}

#ifdef SFA_DEFINE_FINALIZE_KERNEL  // plain (non-template) kernels: defined in exactly one translation unit
// one wave per (quad, job, hand-over between segment s and s+1): the state the later segment assumed against the state
// the earlier one reached
__global__ void __launch_bounds__(256) sdtw_verify_kernel(const DpArgs a, const int n_quads_total) {
    const int64_t w = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int per_quad = a.n_jobs * (a.n_seg - 1);
    if (w >= static_cast<int64_t>(n_quads_total) * per_quad) return;
    const int quad = static_cast<int>(w / per_quad), rest = static_cast<int>(w - static_cast<int64_t>(quad) * per_quad);
    const int job = rest / (a.n_seg - 1), seg = rest - job * (a.n_seg - 1);
    int ci = 0;
    while (ci + 1 < a.n_cls && quad >= a.cls[ci + 1].quad_base) ++ci;
    const int qlen = a.quad_qlen[quad];
    const SegRange next = segment_range(a.job_len[job], qlen, a.n_seg, a.warm_windows, seg + 1);
    if (next.empty) return;
    const int lane = threadIdx.x & 63;
    const float *out = a.verify + verify_slot(a, quad, job, seg, 1) + lane, *in = a.verify + verify_slot(a, quad, job, seg + 1, 0) + lane;
    bool same = true;
    for (int p = 0; p <= a.cls[ci].R; ++p) same = same && (__float_as_uint(out[p * 64]) == __float_as_uint(in[p * 64]));
    if (!__all(same) && lane == 0) a.seg_fail[quad] = 1;
}

// Non-finite query values.  The reference ABORTS on such a read -- a NaN or inf event makes the last row of the cost matrix
// NaN, the window scan (src/sigfish.c:892-899) then returns min_pos = -1, update_aln() traces back from a column far outside
// the matrix and dies in `assert(len >= 0)` (src/sigfish.c:611); observed with the compiled reference for one NaN event, an
// all-NaN query (what z-normalising a zero-variance window gives) and one +inf event: tests/golden/degenerate/.  A batch
// call cannot abort, so these reads are screened out up front: their rows come back valid = 0 (as for a read without
// events: nothing is printed) and sfa_profile_t.non_finite_reads counts them.  One wave per read, coalesced.
__global__ void __launch_bounds__(256) sdtw_screen_kernel(const float *queries, const int64_t *q_off, const int n, uint8_t *bad,
                                                          unsigned *count) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = q_off[i], e = q_off[i + 1];
    bool nf = false;
    for (int64_t j = b + lane; j < e; j += 64) {
        const unsigned u = __float_as_uint(queries[j]);
        nf = nf || ((u & 0x7f800000u) == 0x7f800000u);  // exponent all ones: inf or NaN
    }
    const bool any = __any(nf);
    if (lane == 0) {
        bad[i] = any ? 1 : 0;
        if (any) atomicAdd(count, 1u);
    }
}

__global__ void __launch_bounds__(256) sdtw_finalize_kernel(const FinalizeArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_reads) return;
    const int sl = a.slot_of_read[i];
    const bool strips = a.max_query > 0 && a.q_off[i + 1] - a.q_off[i] > a.max_query;
    if (strips && a.mode != 1) return;
    if (a.mode == 3) {  // fused launch: its pass-2 waves write the rows of every read they handle; the others are skipped reads
        if (sl >= 0) return;
        ResultRow r;
        r.rid = -1;
        r.pos_st = -1;
        r.pos_end = -1;
        r.score = INFINITY;
        r.score2 = INFINITY;
        r.strand = 0;
        r.mapq = 0;
        r.valid = 0;
        r.pad = 0;
        a.out[i] = r;
        return;
    }
    if (a.mode == 2) {  // positions from the traced start column
        if (sl < 0) return;
        ResultRow r = a.out[i];
        if (r.rid < 0) return;
        const int st = a.t_st[i];
        const int end = a.t_st[a.n_reads + i];
        const int rl = a.ref_len[r.rid];
        const int off = a.ref_st_offset[r.rid];
        r.pos_st = ((r.strand == '+') ? st : rl - end) + off;  // src/sigfish.c:971-975
        r.pos_end = ((r.strand == '+') ? end : rl - st) + off;
        a.out[i] = r;
        if (a.span_hist && st >= 0 && end >= st) {
            const int64_t ql = a.q_off[i + 1] - a.q_off[i];
            if (ql > 0) {
                const int64_t b = (static_cast<int64_t>(end - st + 1) * 16) / ql;
                atomicAdd(a.span_hist + (b < kSpanBuckets - 1 ? static_cast<int>(b) : kSpanBuckets - 1), 1u);
            }
        }
        return;
    }
    ResultRow r;
    r.rid = -1;
    r.pos_st = -1;
    r.pos_end = -1;
    r.score = INFINITY;
    r.score2 = INFINITY;
    r.strand = 0;
    r.mapq = 0;
    r.valid = 0;
    r.pad = 0;
    int wjob = -1, wend = -1, wchunk = 0;
    if (sl >= 0 && !a.bad[i]) {
        const int64_t quad = sl >> 2, slot = sl & 3;
        float best = INFINITY, second = INFINITY;
        int end = -1, job = -1;
        for (int ch = 0; ch < a.n_chunks; ++ch) {  // chunks in processing order: a later chunk wins ties
            const int64_t o = (quad * a.n_chunks + ch) * 4 + slot;
            const float b = a.p_best[o], s2 = a.p_second[o];
            const float hi = fmaxf(best, b);
            const float lo2 = fminf(second, s2);
            const bool take = !(b > best);
            second = fminf(hi, lo2);  // second smallest of {best, second, b, s2}
            if (take) {
                best = b;
                end = a.p_end[o];
                job = a.p_job[o];
                wchunk = ch;
            }
        }
        r.valid = 1;
        r.score = best;
        r.score2 = second;
        if (job >= 0) {
            const int rid = a.job_contig[job];
            const int8_t d = a.job_strand[job];
            r.rid = rid;
            r.strand = d;
            r.mapq = mapq_from_scores(best, second);
            wjob = job;
            wend = end;
        }
    }
    if (a.mode == 1) {
        a.w_job[i] = wjob;
        a.w_end[i] = wend;
        a.w_score[i] = r.score;
        a.w_chunk[i] = wchunk;
    }
    if (!strips) a.out[i] = r;
}
#endif

}  // namespace sfa
