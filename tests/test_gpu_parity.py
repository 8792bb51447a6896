"""Parity tests proper: the HIP path through the C-ABI against (a) the golden fixtures produced by the compiled
reference and (b) the oracle on seeded inputs.  Integer fields bit-exact; fp32 scores bit-exact as well (the
north-star tolerance is 1e-4 relative; the kernels are built to hit 0)."""
import numpy as np
import pytest

import sigfish_amd as S
from sigfish_amd import api, synth
from tests.util import case_names, load_case, paf_lines_from_results

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # stated tolerance for d1/d2; asserted at 0 ulp below


def _oracle_ref(O, ref):
    return O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)


def assert_rows_equal(got, want):
    assert np.array_equal(got["valid"], want["valid"])
    v = want["valid"] == 1  # skipped reads (no events) carry no alignment; the reference prints nothing for them
    for f in ("rid", "strand", "pos_st", "pos_end", "mapq"):
        assert np.array_equal(got[f][v], want[f][v]), (f, np.nonzero(got[f] != want[f])[0][:10], got[f][:8], want[f][:8])
    assert np.allclose(got["score"][v], want["score"][v], rtol=REL_TOL)
    assert np.array_equal(got["score"][v].view(np.uint32), want["score"][v].view(np.uint32))
    assert np.array_equal(got["score2"][v].view(np.uint32), want["score2"][v].view(np.uint32))


# Every setting of the options that selects another kernel or another hand-over between pass 1 and pass 2 (include/sigfish_amd.h:
# rows never depend on options), most pinned to the throughput shapes (16 lanes per read):
#   two_pass      the defaults at these batch sizes: pass 2 as its own launch, LDS checkpoints where the shapes allow
#   fused         pass 2 by ticket inside the fill launch whatever the batch size: the LDS-checkpoint fill (<= 16 rows per lane,
#                 up to 1024 events) and the 32-row fill with its snapshots in HBM
#   unfused       ... and never
#   hbm_ckpt      every snapshot of pass 1 in HBM
#   lds_backoff   LDS checkpoints with no head start at all, so that pass 2 must back off to the sparse store / the strand start
#   dense_ckpt    snapshots every 32 steps and no head start: every read backs off through several of them
#   strips_backoff  the same for the row strips of queries beyond 2048 events
#   wide2 / wide4_dense / lds_wide4_backoff  the small-batch shapes (rows per lane / 2 and / 4)
#   auto          what the planner picks by itself for these small batches (lane widening, column segments)
MODES = {"two_pass": {"lane_widening": 1},
         "fused": {"lds_ckpt": 2, "fused_trace": 2, "lane_widening": 1},
         "unfused": {"lds_ckpt": 2, "fused_trace": 0, "lane_widening": 1},
         "hbm_ckpt": {"lds_ckpt": 0, "lane_widening": 1},
         "hbm_fused": {"lds_ckpt": 0, "fused_trace": 2, "lane_widening": 1},  # (only the 32-row fill carries tickets without LDS checkpoints)
         "lds_backoff": {"trace_margin": 0, "lane_widening": 1, "lds_ckpt": 2, "fused_trace": 2},
         "lds_wide4_backoff": {"trace_margin": 0, "lane_widening": 4, "column_segments": 1, "lds_ckpt": 2, "fused_trace": 2},
         "strips_backoff": {"trace_margin": 0, "ckpt_interval": 64},  # every strip of pass 2 backs off
         "dense_ckpt": {"ckpt_interval": 32, "trace_margin": 0, "lane_widening": 1},
         "wide2": {"lane_widening": 2}, "wide4_dense": {"lane_widening": 4, "ckpt_interval": 32, "trace_margin": 0},
         "auto": {}}


def _aligner(ref, flag, mode="two_pass"):
    al = S.Aligner(ref, flag)
    for k, v in MODES[mode].items():
        al.set_option(k, v)
    return al


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", case_names())
def test_golden_cases(oracle, name, mode):
    """Fixture inputs (real reads, reference-built event arrays) -> rows and PAF text of the compiled reference."""
    c = load_case(name)
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    with _aligner(ref, c["flag"], mode) as al:
        res = al.align_db(c["queries"], c["q_off"])
    for f in ("rid", "pos_st", "pos_end", "mapq", "strand"):
        assert np.array_equal(res[f], c[f]), (f, res[f], c[f])
    assert np.array_equal(res["score"].view(np.uint32), c["score"].view(np.uint32))
    assert np.array_equal(res["score2"].view(np.uint32), c["score2"].view(np.uint32))
    if not c["sam"]:
        lines = []
        vi = 0
        for i, rid in enumerate(c["read_ids"]):
            if not c["read_valid"][i]:
                continue
            r = res[vi]
            end_raw = int(c["ev_start_last"][vi]) + int(c["ev_len_last"][vi])
            lines.append(S.paf_row(r, rid, ref.names[int(r["rid"])], int(c["ev_start_first"][vi]), end_raw,
                                   int(c["qend"][i]) - 1 - int(c["qstart"][i]), int(c["len_raw"][i]),
                                   int(ref.seq_lengths[int(r["rid"])])))
            vi += 1
        assert "".join(lines) == c["out_text"]


@pytest.mark.parametrize("wl,n", [("ncov_r9_dna_q250", 203), ("sequin_r9_rna_q250", 64),
                                  ("rna004_fullref_dtwstd_q250", 48)])
def test_synthetic_vs_oracle(oracle, wl, n):
    ref, flag, q, q_off, meta = synth.workload(wl, n_reads=n, seed=11)
    with S.Aligner(ref, flag) as al:
        got = al.align_db(q, q_off)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16)
    assert_rows_equal(got, want)


def _small_ref(rng, lens, rna, quant=False):
    def arr(n):
        a = (rng.integers(-6, 7, n) / 4).astype(np.float32) if quant else rng.normal(size=n).astype(np.float32)
        return a
    fw = [arr(n) for n in lens]
    rv = None if rna else [arr(n) for n in lens]
    names = [f"c{i}" for i in range(len(lens))]
    return S.RefModel(names, [n + 5 for n in lens], lens, rng.integers(0, 3, len(lens)) if rna else [0] * len(lens),
                      fw, rv)


@pytest.mark.parametrize("gpu_mode", list(MODES))
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("mode", ["dna", "rna", "rna_std", "rna_inv"])
def test_ragged_and_ties(oracle, seed, mode, gpu_mode):
    """Edge cases the reference's own runs exercise: ragged query lengths (incl. 1..24 and >256), empty reads,
    contigs shorter than the query, exact ties from quantised levels, batches that do not fill a wavefront."""
    rng = np.random.default_rng(1000 + seed)
    rna = mode != "dna"
    flag = {"dna": 0, "rna": S.RNA, "rna_std": S.RNA | S.DTW, "rna_inv": S.RNA | S.INV}[mode]
    quant = seed % 2 == 0
    lens = [int(x) for x in rng.integers(3, 900, size=int(rng.integers(1, 9)))]
    if seed == 3:
        lens = [1, 2, 700]
    ref = _small_ref(rng, lens, rna, quant)
    n = int(rng.integers(1, 40))
    qlens = rng.choice([0, 1, 2, 7, 25, 63, 64, 65, 100, 128, 129, 250, 256, 257, 300, 512], size=n)
    if seed == 5:
        qlens[:] = 250
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = (rng.integers(-6, 7, int(q_off[-1])) / 4).astype(np.float32) if quant else rng.normal(size=int(q_off[-1])).astype(np.float32)
    with _aligner(ref, flag, gpu_mode) as al:
        got = al.align_db(q, q_off)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=8)
    assert_rows_equal(got, want)


@pytest.mark.parametrize("gpu_mode", list(MODES))
@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("mode", ["dna", "rna", "rna_std", "rna_inv"])
def test_long_queries(oracle, seed, mode, gpu_mode):
    """`-q` beyond 512 events: 32 lanes per read up to 1024 events, a whole wave per read up to 2048, mixed in one
    batch with the ordinary classes; contigs shorter than the query included."""
    rng = np.random.default_rng(7000 + seed)
    rna = mode != "dna"
    flag = {"dna": 0, "rna": S.RNA, "rna_std": S.RNA | S.DTW, "rna_inv": S.RNA | S.INV}[mode]
    quant = seed == 1
    lens = [int(x) for x in rng.integers(600, 6000, size=int(rng.integers(1, 5)))] + [int(rng.integers(3, 500))]
    ref = _small_ref(rng, lens, rna, quant)
    qlens = rng.choice([0, 64, 250, 512, 513, 600, 1000, 1023, 1024, 1025, 1500, 2047, 2048], size=int(rng.integers(3, 20)))
    qlens[0] = [2048, 1024, 1025][seed]
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = (rng.integers(-6, 7, int(q_off[-1])) / 4).astype(np.float32) if quant else rng.normal(size=int(q_off[-1])).astype(np.float32)
    with _aligner(ref, flag, gpu_mode) as al:
        got = al.align_db(q, q_off)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=8)
    assert_rows_equal(got, want)


@pytest.mark.parametrize("gpu_mode", ["two_pass", "fused", "dense_ckpt", "auto", "strips_backoff"])
@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("mode", ["dna", "rna", "rna_std", "rna_inv"])
def test_row_strips(oracle, seed, mode, gpu_mode):
    """`-q` beyond 2048 events (the reference has no limit): row strips of 2048 query rows, one wave per (read, contig,
    strand), mixed in one batch with every ordinary class; strip edges (2049, 4096, 4097), contigs shorter than the query
    and shorter than a wave, exact ties from quantised levels."""
    rng = np.random.default_rng(9000 + seed)
    rna = mode != "dna"
    flag = {"dna": 0, "rna": S.RNA, "rna_std": S.RNA | S.DTW, "rna_inv": S.RNA | S.INV}[mode]
    quant = seed == 1
    lens = [int(x) for x in rng.integers(600, 7000, size=int(rng.integers(1, 4)))] + [int(rng.integers(3, 60)), int(rng.integers(60, 500))]
    ref = _small_ref(rng, lens, rna, quant)
    # (2560 / 3072 / 3584 / 5376: the last lengths of two / three strips of 64 x 20 / 24 / 28 rows)
    qlens = rng.choice([0, 64, 250, 1000, 2048, 2049, 2100, 2560, 2561, 3000, 3072, 3073, 3584, 3585, 4095, 4096, 4097, 5000, 5376, 5377, 6500],
                       size=int(rng.integers(3, 12)))
    qlens[0] = [2049, 4096, 4097, 6500][seed]
    if seed == 3:
        qlens = qlens[qlens > 2048]  # a batch of long reads only
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = (rng.integers(-6, 7, int(q_off[-1])) / 4).astype(np.float32) if quant else rng.normal(size=int(q_off[-1])).astype(np.float32)
    with _aligner(ref, flag, gpu_mode) as al:
        got = al.align_db(q, q_off)
        again = al.align_db(q, q_off)  # buffers reused
    assert got.tobytes() == again.tobytes()
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=8)
    assert_rows_equal(got, want)


def test_row_strips_ncov_and_groups(oracle):
    """q = 2500 and 5000 against the nCoV reference (both strands): noisy copies of reference stretches, so that real
    alignments exist; then the same batch with a boundary-row budget that forces one read per launch."""
    ref, flag, q250, off250, meta = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=11)
    rng = np.random.default_rng(13)
    qlens = np.array([2500, 250, 5000, 2049, 1000])
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = np.empty(int(q_off[-1]), np.float32)
    fw, rv = ref.forward[0], ref.reverse[0]
    for i, l in enumerate(qlens):
        src = fw if i % 2 == 0 else rv
        st = int(rng.integers(0, len(src) - l))
        seg = src[st:st + l] + rng.normal(scale=0.3, size=l).astype(np.float32)
        q[q_off[i]:q_off[i + 1]] = ((seg - seg.mean()) / seg.std()).astype(np.float32)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16)
    with S.Aligner(ref, flag) as al:
        got = al.align_db(q, q_off)
        assert_rows_equal(got, want)
        assert (got["mapq"][[0, 2, 3]] > 0).all() and [chr(c) for c in got["strand"][[0, 2, 3]]] == ["+", "+", "-"]
        for interval, margin in ((64, 0), (4, 3), (4096, -1), (512, 100)):  # pass 2 from any checkpoint spacing, incl. the back-off
            al.set_option("ckpt_interval", interval)
            al.set_option("trace_margin", margin)
            assert al.align_db(q, q_off).tobytes() == got.tobytes()
        al.set_option("ckpt_interval", 0)
        al.set_option("trace_margin", -1)
        al.set_option("ckpt_budget_bytes", 1 << 20)  # less than one read's boundary rows: groups of one read
        assert al.align_db(q, q_off).tobytes() == got.tobytes()
        assert al.profile()["fill_launches"] == 1 + 3
        al.set_option("ckpt_budget_bytes", 16 << 30)
        for interval, margin in ((64, 0), (4, 3), (4096, -1), (512, 100)):  # ... and with all reads in one launch
            al.set_option("ckpt_interval", interval)
            al.set_option("trace_margin", margin)
            assert al.align_db(q, q_off).tobytes() == got.tobytes()


def test_row_strips_extreme_lengths(oracle):
    """Queries of 12 345 and 20 000 events and one as long as the reference itself (29 898 events: 15 strips, a single
    window per strand) next to ordinary ones."""
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=0)
    rng = np.random.default_rng(5)
    qlens = np.array([20000, 12345, 2049, 250, 29898])
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = np.empty(int(q_off[-1]), np.float32)
    for i, l in enumerate(qlens):
        src = ref.forward[0] if i % 2 == 0 else ref.reverse[0]
        st = int(rng.integers(0, len(src) - l + 1))
        seg = src[st:st + l] + rng.normal(scale=0.3, size=l).astype(np.float32)
        q[q_off[i]:q_off[i + 1]] = ((seg - seg.mean()) / seg.std()).astype(np.float32)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=5)
    with S.Aligner(ref, flag) as al:
        got = al.align_db(q, q_off)
    assert_rows_equal(got, want)
    assert [chr(c) for c in got["strand"]] == ["+", "-", "+", "-", "+"] and (got["mapq"] == 60).all()


def test_long_queries_ncov(oracle):
    """q = 1000 and 2000 against the nCoV reference (both strands), default checkpointing."""
    ref, flag, q250, off250, meta = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=11)
    rng = np.random.default_rng(12)
    qlens = np.array([1000, 2000, 1000, 700, 250, 2000, 1000])
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = np.empty(int(q_off[-1]), np.float32)
    fw = ref.forward[0]
    for i, l in enumerate(qlens):  # noisy copies of reference stretches, so that real alignments exist
        st = int(rng.integers(0, len(fw) - l))
        seg = fw[st:st + l] + rng.normal(scale=0.3, size=l).astype(np.float32)
        q[q_off[i]:q_off[i + 1]] = ((seg - seg.mean()) / seg.std()).astype(np.float32)
    with S.Aligner(ref, flag) as al:
        got = al.align_db(q, q_off)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16)
    assert_rows_equal(got, want)
    assert (got["mapq"][[0, 1, 2, 5, 6]] > 0).all()


@pytest.mark.parametrize("interval,margin", [(4, 0), (8, 3), (64, 0), (256, 100), (1024, -1)])
def test_checkpoint_intervals(oracle, interval, margin):
    """Pass 2 must recover the same start column from any checkpoint spacing, including the back-off path
    (margin 0 starts right at the winner, so the path almost always begins before the first checkpoint tried)."""
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=37, seed=5)
    with S.Aligner(ref, flag) as al:
        al.set_option("ckpt_interval", interval)
        al.set_option("trace_margin", margin)
        got = al.align_db(q, q_off)
        prof = al.profile()
    assert prof["ckpt_interval"] == interval
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16)
    assert_rows_equal(got, want)


def test_checkpoint_budget_picks_larger_interval():
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=64, seed=6)
    with S.Aligner(ref, flag) as al:
        a = al.align_db(q, q_off)
        pa = al.profile()
        al.set_option("ckpt_budget_bytes", 64 * 1024)
        b = al.align_db(q, q_off)
        pb = al.profile()
    assert pa["ckpt_interval"] == 512 and pb["ckpt_interval"] > 512 and pb["ckpt_bytes"] <= 64 * 1024
    assert a.tobytes() == b.tobytes()


def test_huge_batch_is_sliced(oracle):
    """A batch whose checkpoints exceed the budget at T = 512 is cut into slices of contiguous reads; rows unchanged."""
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=300, seed=31)
    with S.Aligner(ref, flag) as al:
        whole = al.align_db(q, q_off)
        p0 = al.profile()
        al.set_option("ckpt_budget_bytes", 8 << 20)   # 300 reads need ~40 MB at T = 512
        al.set_option("min_slice_reads", 40)
        sliced = al.align_db(q, q_off)
        p1 = al.profile()
    assert sliced.tobytes() == whole.tobytes()
    assert p1["fill_launches"] > 1 and p0["fill_launches"] == 1 and p1["cells"] == p0["cells"]
    assert p1["ckpt_interval"] == 512 and p1["ckpt_bytes"] <= 8 << 20
    assert_rows_equal(whole, oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16))


def test_long_reference_1mb(oracle):
    """BASELINE config 4 shape (1 Mb reference, k=9): the oracle needs a 1 GB matrix per read and strand, so three
    reads; the GPU uses a larger checkpoint interval only if the budget says so (here it does not)."""
    ref, flag, q, q_off, meta = synth.workload("r10_dna_1mb_q250", n_reads=3, seed=2)
    with S.Aligner(ref, flag) as al:
        got = al.align_db(q, q_off)
        al.set_option("ckpt_budget_bytes", 1 << 20)   # force a large interval (few checkpoints)
        got2 = al.align_db(q, q_off)
        p2 = al.profile()
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=3)
    assert_rows_equal(got, want)
    assert_rows_equal(got2, want)
    assert p2["ckpt_interval"] > 1024


def test_many_batches_reuse_context(oracle):
    """Buffers only grow; alternating big/small/empty batches through one context must not leak state."""
    ref, flag, q, q_off, meta = synth.workload("sequin_r9_rna_q250", n_reads=40, seed=9)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=8)
    with S.Aligner(ref, flag) as al:
        for lo, hi in ((0, 40), (3, 4), (10, 10), (0, 17), (39, 40), (5, 33)):
            got = al.align_db(q[q_off[lo]:q_off[hi]], q_off[lo:hi + 1] - q_off[lo])
            assert got.tobytes() == want[lo:hi].tobytes()


def test_submit_wait_pair(oracle):
    """sfa_submit_batch / sfa_wait_batch: the asynchronous half-calls return the rows of the synchronous call."""
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=64, seed=21)
    with S.Aligner(ref, flag) as al:
        want = al.align_db(q, q_off)
        al.submit(q, q_off)
        host_work = np.cumsum(np.arange(100000))  # anything: the batch is on the GPU meanwhile
        got = al.wait()
        assert got.tobytes() == want.tobytes() and host_work[-1] > 0
        with pytest.raises(S.SfaError, match="no batch"):
            al.wait()
        # a second submit before wait discards the first batch
        al.submit(q[:q_off[8]], q_off[:9])
        al.submit(q, q_off)
        assert al.wait().tobytes() == want.tobytes()
        # mismatched size
        al.submit(q, q_off)
        out = np.zeros(3, S.api.RESULT_DTYPE)
        import ctypes as C
        assert al._L.sfa_wait_batch(al._h, out.ctypes.data_as(C.c_void_p), 3) != 0
        # empty batch
        al.submit(np.zeros(0, np.float32), np.zeros(1, np.int64))
        assert len(al.wait()) == 0
    assert_rows_equal(want, oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=8))


def test_align_events_entry(oracle):
    """align_db-shaped entry: AoS event tables + qstart/qend, as db_t holds them."""
    from sigfish_amd.api import EVENT_DTYPE
    rng = np.random.default_rng(5)
    ref = _small_ref(rng, [400, 90], False)
    tables, qs, qe, flat = [], [], [], []
    for i in range(9):
        ne = int(rng.integers(0, 400)) if i != 4 else 0
        t = np.zeros(ne, EVENT_DTYPE)
        t["mean"] = rng.normal(size=ne)
        t["start"] = np.arange(ne) * 7
        t["length"] = 7
        a = min(50, ne)
        b = min(a + 250, ne)
        tables.append(t if ne else None)
        qs.append(a)
        qe.append(b)
        flat.append(t["mean"][a:b].astype(np.float32) if ne else np.zeros(0, np.float32))
    with S.Aligner(ref, 0) as al:
        got = al.align_events(tables, qs, qe)
    q_off = np.concatenate([[0], np.cumsum([len(f) for f in flat])]).astype(np.int64)
    want = oracle.align_batch(np.concatenate(flat), q_off, _oracle_ref(oracle, ref), 0, threads=4)
    assert_rows_equal(got, want)


def test_full_size_properties():
    """BASELINE config 3 at full size (100k reads x nCoV): size-independent properties instead of the oracle.
    (1) determinism / idempotence: same batch twice -> identical rows; (2) permutation equivariance: results
    follow their reads; (3) sub-batch consistency: a slice aligned alone equals the slice of the big batch;
    (4) truth recovery: reads drawn from the reference map back onto their origin."""
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=100_000, seed=3)
    n = meta["n_reads"]
    with S.Aligner(ref, flag) as al:
        a = al.align_db(q, q_off)
        b = al.align_db(q, q_off)
        assert a.tobytes() == b.tobytes()
        # reversed read order
        lens = (q_off[1:] - q_off[:-1])
        order = np.arange(n)[::-1]
        q2 = np.concatenate([q[q_off[i]:q_off[i + 1]] for i in order[:5000]])
        qo2 = np.concatenate([[0], np.cumsum(lens[order[:5000]])]).astype(np.int64)
        c = al.align_db(q2, qo2)
        assert c.tobytes() == a[order[:5000]].tobytes()
        d = al.align_db(q[q_off[777]:q_off[1301]], q_off[777:1302] - q_off[777])
        assert d.tobytes() == a[777:1301].tobytes()
    assert (a["valid"] == 1).all() and (a["mapq"] <= 60).all()
    tr = meta["truth"]
    strand_ok = (a["strand"] == np.where(tr["strand"] == 0, ord("+"), ord("-")))
    rl = int(ref.ref_lengths[0])
    exp_st = np.where(tr["strand"] == 0, tr["start"], rl - (tr["start"] + tr["span"]))
    near = np.abs(a["pos_st"] - exp_st) <= 30
    full = lens == 250
    assert (strand_ok & near)[full].mean() > 0.97
    assert (a["pos_end"] > a["pos_st"]).all()
    assert (a["score2"] >= a["score"]).all()


def test_bad_query_window_is_rejected():
    from sigfish_amd.api import EVENT_DTYPE
    ref = _small_ref(np.random.default_rng(0), [50], False)
    t = np.zeros(30, EVENT_DTYPE)
    with S.Aligner(ref, 0) as al:
        with pytest.raises(S.SfaError, match="outside"):
            al.align_events([t], [10], [31])
        with pytest.raises(S.SfaError, match="outside"):
            al.align_events([t], [-1], [20])


def test_context_lifecycle_does_not_leak():
    """init / align / destroy in a loop: device memory returns to where it started."""
    import ctypes as C
    from sigfish_amd import _lib

    def free_bytes():
        f, t = C.c_uint64(), C.c_uint64()
        assert _lib.load().sfa_device_memory(0, C.byref(f), C.byref(t)) == 0
        return f.value

    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=64, seed=3)
    with S.Aligner(ref, flag) as al:
        want = al.align_db(q, q_off)
    free0 = free_bytes()
    for _ in range(40):
        with S.Aligner(ref, flag) as al:
            assert al.align_db(q, q_off).tobytes() == want.tobytes()
    free1 = free_bytes()
    assert abs(free0 - free1) < 64 << 20, (free0, free1)


def test_column_segments_and_their_fallback(oracle):
    """Small batches cut every strand into segments that start from a guessed state and are accepted only when the state
    at the hand-over equals the state the previous segment reached.  With the normal warm-up they verify; with no warm-up
    at all they cannot, and the batch is walked again unsegmented -- same rows either way."""
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=96, seed=17)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16)
    with S.Aligner(ref, flag) as al:
        rows = al.align_db(q, q_off)
        p = al.profile()
        assert p["n_segments"] >= 3 and p["segment_reruns"] == 0 and p["n_chunks"] == 2 * p["n_segments"]
        assert_rows_equal(rows, want)
        for seg in (2, 5, 16, 64):
            al.set_option("column_segments", seg)
            assert al.align_db(q, q_off).tobytes() == rows.tobytes() and al.profile()["segment_reruns"] == 0
        al.set_option("column_segments", 1)
        assert al.align_db(q, q_off).tobytes() == rows.tobytes() and al.profile()["n_segments"] == 1
        al.set_option("column_segments", 6)
        al.set_option("segment_warm_windows", 0)      # the guess is used as is: the hand-over check must catch it
        assert al.align_db(q, q_off).tobytes() == rows.tobytes()
        assert al.profile()["segment_reruns"] == 1
        al.set_option("ckpt_interval", 64)              # dense checkpoints across segment boundaries, pass 2 from any of them
        al.set_option("segment_warm_windows", 4)
        al.set_option("trace_margin", 0)
        assert al.align_db(q, q_off).tobytes() == rows.tobytes()


@pytest.mark.parametrize("seed", range(4))
def test_column_segments_on_tie_heavy_data(oracle, seed):
    """Quantised levels (exact ties everywhere), ragged query lengths, a short warm-up: whatever verifies is right,
    whatever does not is re-run."""
    rng = np.random.default_rng(300 + seed)
    lens = [int(x) for x in rng.integers(3000, 9000, size=2)]
    ref = _small_ref(rng, lens, False, quant=True)
    qlens = rng.choice([0, 7, 25, 63, 64, 65, 100, 128, 129, 250, 256, 300, 512], size=40)
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = (rng.integers(-6, 7, int(q_off[-1])) / 4).astype(np.float32)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), 0, threads=8)
    with S.Aligner(ref, 0) as al:
        for seg, warm in ((0, 4), (4, 1), (8, 2), (3, 0)):
            al.set_option("column_segments", seg)
            al.set_option("segment_warm_windows", warm)
            assert_rows_equal(al.align_db(q, q_off), want)


def test_degenerate_and_non_finite_queries_against_the_reference():
    """tests/golden/degenerate (oracle/make_golden_degenerate.py, answered by the compiled reference's own align_db):
    constant, quantised and two-level queries -- the reference prints rows (exact ties: score == score2, mapq 0), and so
    must we, bit for bit; a read with a NaN or inf event -- the reference ABORTS (assert in update_aln, src/sigfish.c:611:
    reference_on_non_finite.txt), a batch call cannot, so that read comes back valid = 0 and is counted, and the other
    reads of the batch are unaffected."""
    import os
    from tests.util import GOLD
    z = np.load(os.path.join(GOLD, "degenerate", "degenerate.npz"))
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=16, seed=2)
    assert "Assertion `len >= 0' failed" in open(os.path.join(GOLD, "degenerate", "reference_on_non_finite.txt")).read()
    for mode in ("two_pass", "fused", "wide4_dense"):
        with _aligner(ref, flag, mode) as al:
            assert_rows_equal(al.align_db(z["queries_finite"], z["q_off"]), z["rows_finite"].view(S.RESULT_DTYPE))
            assert al.profile()["non_finite_reads"] == 0
            got = al.align_db(z["queries"], z["q_off"])
            assert al.profile()["non_finite_reads"] == len(z["non_finite"])
        assert got.tobytes() == z["rows"].view(S.RESULT_DTYPE).tobytes()
        assert list(np.flatnonzero(got["valid"] == 0)) == list(z["non_finite"])


@pytest.mark.parametrize("fused", [1, 0])
def test_backoff_to_the_sparse_checkpoints_inside_the_fused_launch(oracle, fused):
    """A strand long enough for the sparse HBM checkpoints behind the LDS ones (every 32 768 steps) and no head start at all for
    pass 2 (trace_margin 0): most winners beyond column 32 768 start pass 2 from the saved snapshot, fail, and back off to
    the sparse store -- which, in the fused launch, another XCD wrote during the same launch.  (Found by the fuzz campaign
    while those stores were not yet write-through: seed 2026, iteration 8414.)"""
    rng = np.random.default_rng(8414)
    ref = _small_ref(rng, [70000], True, quant=True)
    qlens = rng.integers(20, 257, size=120)
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = (rng.integers(-8, 9, int(q_off[-1])) / 4).astype(np.float32)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), S.RNA | S.INV, threads=8)
    assert (want["pos_end"] > 33000).sum() > 20
    with S.Aligner(ref, S.RNA | S.INV) as al:
        for k, v in (("trace_margin", 0), ("lane_widening", 1), ("lds_ckpt", 2), ("fused_trace", 2 * fused), ("prio_unit", 64)):
            al.set_option(k, v)
        for _ in range(3):
            assert_rows_equal(al.align_db(q, q_off), want)
        assert al.profile()["lds_ckpt"] == (2 if fused else 1)


def test_non_finite_long_query_is_skipped_too():
    """The same screen in front of the row-strip path (queries beyond 2048 events)."""
    rng = np.random.default_rng(5)
    ref = _small_ref(rng, [700, 400], False)
    lens = [2500, 40, 2300]
    q = rng.normal(size=sum(lens)).astype(np.float32)
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    with S.Aligner(ref, 0, device=0) as al:
        clean = al.align_db(q, q_off)
        q[100] = np.nan
        got = al.align_db(q, q_off)
        assert al.profile()["non_finite_reads"] == 1
    assert got["valid"].tolist() == [0, 1, 1] and got["rid"][0] == -1 and got[1:].tobytes() == clean[1:].tobytes()


def test_options_are_validated():
    ref = _small_ref(np.random.default_rng(0), [50], False)
    with S.Aligner(ref, 0) as al:
        for key, bad in (("ckpt_interval", 3), ("ckpt_interval", 48), ("ckpt_budget_bytes", -1), ("waves_per_simd", 0),
                         ("waves_per_simd", 9), ("lane_widening", 3), ("widen_below", -1), ("min_slice_reads", 0),
                         ("no_such_option", 1)):
            with pytest.raises(S.SfaError):
                al.set_option(key, bad)
        for key, ok in (("ckpt_interval", 0), ("ckpt_interval", 64), ("trace_margin", -1), ("trace_margin", 0),
                        ("lane_widening", 0), ("widen_below", 5), ("ev_parallel", 0), ("ev_parallel", 3)):
            al.set_option(key, ok)
        # options of earlier rounds that named rejected variants are gone, not silently accepted
        for key in ("single_pass", "strip_pipeline", "strip_chain", "balanced_strips", "long_overlap", "mixed_quads", "adaptive_margin"):
            with pytest.raises(S.SfaError, match="unknown option"):
                al.set_option(key, 1)
        # the test hooks are not options: refused unless the process asks for them (tests/test_bounded_waits_gpu.py does)
        import os
        if os.environ.get("SFA_TEST_HOOKS") != "1":
            with pytest.raises(S.SfaError, match="test hook"):
                al.set_option("debug_drop_quad", 0)


def test_no_device_fallback_is_loud():
    ref = _small_ref(np.random.default_rng(0), [50], False)
    with pytest.raises(S.SfaError):
        S.Aligner(ref, 0, device=99)


def test_pass_2_head_start_follows_the_previous_batch(oracle):
    """Queries beyond 256 events keep their snapshots in HBM; there pass 2 starts a whole query length (+ lanes) in front of
    the winning cell for the first batch of a context and, from then on, as far as 99.9 % of the previous batch's alignments
    spanned (event detection over-segments: ~2/3 of a column per event) -- rows never change, a path that is longer backs off."""
    ref, flag, q, q_off, _ = synth.workload("ncov_r9_dna_q500", n_reads=300, seed=5)
    want = oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=16)
    with S.Aligner(ref, flag) as al:
        al.set_option("lane_widening", 1)
        first = al.align_db(q, q_off)
        m1 = al.profile()["trace_margin"]
        second = al.align_db(q, q_off)
        m2 = al.profile()["trace_margin"]
        assert m1 == 500 + 16 and 500 * 10 // 16 <= m2 < m1, (m1, m2)
        assert_rows_equal(first, want)
        assert_rows_equal(second, want)
        # reads whose paths are LONGER than the learnt head start (a query stretched over twice as many columns) still come out right
        rng = np.random.default_rng(9)
        src = ref.forward[0]
        qs, offs = [], [0]
        for _ in range(80):
            st = int(rng.integers(0, len(src) - 1100))
            seg = src[st:st + 1000:2] + rng.normal(scale=0.2, size=500).astype(np.float32)  # every second level: ~2 columns per event
            qs.append(((seg - seg.mean()) / seg.std()).astype(np.float32))
            offs.append(offs[-1] + 500)
        q2, off2 = np.concatenate(qs), np.array(offs, np.int64)
        got = al.align_db(q2, off2)
        assert al.profile()["trace_margin"] == m2  # planned with what the batch before had shown
        assert_rows_equal(got, oracle.align_batch(q2, off2, _oracle_ref(oracle, ref), flag, threads=16))
        al.align_db(q, q_off)
        assert 500 * 10 // 16 <= al.profile()["trace_margin"] <= 500 + 16  # (whatever those paths spanned: never beyond a whole query)
        al.set_option("trace_margin", 500 + 16)  # (pinned: a whole query length + lanes, as for a context's first batch)
        assert al.align_db(q, q_off).tobytes() == first.tobytes() and al.profile()["trace_margin"] == 500 + 16
        # the same with pass 2 inside the fill launch: its waves keep the histogram of spans themselves
        al.set_option("trace_margin", -1)
        assert al.profile()["fused_trace"] == 0  # (300 reads: fewer wave-tasks than wave slots, pass 2 as its own launch)
        al.set_option("fused_trace", 2)
        assert al.align_db(q, q_off).tobytes() == first.tobytes() and al.profile()["fused_trace"] == 1 and al.profile()["trace_ms"] == 0
        assert al.align_db(q, q_off).tobytes() == first.tobytes() and 500 * 10 // 16 <= al.profile()["trace_margin"] < 500 + 16


def test_reference_of_32_megabases_against_its_cut_outs():
    """Maximum sizes, reference: a 32 Mb contig (3.2 x 10^7 columns per strand, 12 GB of checkpoint offsets beyond 32 bits, column
    segments for the small batch).  No CPU implementation fills a 250 x 3.2 x 10^7 matrix per read and strand in test time, so a
    size-independent property: the best alignment of a read against the whole contig is, bit for bit, its best alignment against
    a 500 000-column cut-out around it (cut at a multiple of the query length, so that the windows of src/sigfish.c:891-901
    coincide), positions shifted by the cut.  (Reads are NOT expected at their origin here: the reference z-normalises a contig
    with fp32 accumulators, src/genref.c:23-47, whose sum stops growing near 2 x 10^9 -- beyond ~1.6 x 10^7 k-mers of level ~90
    the mean is wrong for the reference and for this build alike; tests/campaigns/big_reference.py prints it for 4 .. 128 Mb.)"""
    lv = synth.kmer_levels(6, 1)
    ref = api.RefModel.from_records([("big", synth.random_sequence(32_000_000, 11))], lv, 6, 0, 250)
    rl = int(ref.ref_lengths[0])
    n, W = 64, 250 * 2000
    q, q_off, _ = synth.make_reads(ref, n, qlen=250, seed=5, short_frac=0.0)
    with S.Aligner(ref, 0) as al:
        a = al.align_db(q, q_off)
        assert al.align_db(q, q_off).tobytes() == a.tobytes()
    assert (a["valid"] == 1).all()
    for i in range(0, n, 8):
        fwd = a["strand"][i] == ord("+")
        own = int(a["pos_st"][i]) if fwd else rl - int(a["pos_end"][i])  # column of the hit in its strand's own array
        lo = max(0, (own - W // 2) // 250 * 250)
        hi = min(rl, lo + W)
        f, r = ref.forward[0], ref.reverse[0]
        sub_f, sub_r = (f[lo:hi], r[rl - hi:rl - lo]) if fwd else (f[rl - hi:rl - lo], r[lo:hi])
        sub = api.RefModel(["sub"], [hi - lo + 5], [hi - lo], [0], [np.ascontiguousarray(sub_f)], [np.ascontiguousarray(sub_r)])
        with S.Aligner(sub, 0) as al2:
            c = al2.align_db(q[q_off[i]:q_off[i + 1]], np.array([0, 250], np.int64))
        shift = lo if fwd else rl - hi  # positions are reported on the forward strand
        assert c["score"][0].tobytes() == a["score"][i].tobytes() and c["strand"][0] == a["strand"][i], (i, a[i], c[0])
        assert int(c["pos_st"][0]) + shift == int(a["pos_st"][i]) and int(c["pos_end"][0]) + shift == int(a["pos_end"][i]), (i, a[i], c[0], lo)


def test_batch_of_a_million_reads():
    """Maximum sizes, batch: 1 000 000 reads in one call (ten times the headline batch; 1 GB of queries), built from 50 000
    reads repeated twenty times: every repetition must give the rows of the first, which are the rows of the 50 000 alone."""
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=50_000, seed=17)
    reps = 20
    lens = q_off[1:] - q_off[:-1]
    qq = np.tile(q, reps)
    qo = np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.int64)
    with S.Aligner(ref, flag) as al:
        unit = al.align_db(q, q_off)
        big = al.align_db(qq, qo)
    assert len(big) == reps * len(unit)
    for k in range(reps):
        assert big[k * len(unit):(k + 1) * len(unit)].tobytes() == unit.tobytes(), k
