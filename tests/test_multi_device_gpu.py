"""Multi-GPU behind the C-ABI: sfa_init_devices shards every batch over the listed devices in
contiguous read ranges and returns the rows in input order.  On a one-GPU box the device is listed two and three times
(several shards on one GPU, each with its own stream, scratch and host thread); on a box with G > 1 GPUs the same tests
also run over [0 .. G-1], its reverse and [0, 1] (tests/util.py: device_lists) -- distinct devices, peer copies of the
reference model, one host thread per device -- without a code change.  Rows must be identical to the single-device
context's, through every host-buffer entry point."""
import numpy as np
import pytest

import sigfish_amd as S
from sigfish_amd import synth
from tests.util import device_lists, distinct_device_list

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wl,n", [("ncov_r9_dna_q250", 301), ("sequin_r9_rna_q250", 97), ("rna004_fullref_dtwstd_q250", 23)])
@pytest.mark.parametrize("devices", device_lists(), ids=lambda d: "dev" + "_".join(map(str, d)))
def test_sharded_rows_equal_single_device_rows(wl, n, devices, oracle):
    ref, flag, q, q_off, _ = synth.workload(wl, n_reads=n, seed=3)
    q_off = q_off.copy()
    with S.Aligner(ref, flag, device=0) as one, S.Aligner(ref, flag, devices=devices) as many:
        assert one.n_devices() == 1 and many.n_devices() == len(devices)
        want = one.align_db(q, q_off)
        got = many.align_db(q, q_off)
        assert got.tobytes() == want.tobytes()
        many.submit(q, q_off)
        assert many.wait().tobytes() == want.tobytes()
        p = many.profile()
        assert p["cells"] == one.profile()["cells"] and p["fill_ms"] > 0
        # fewer reads than shards, and an empty batch
        assert many.align_db(q[:q_off[1]], q_off[:2]).tobytes() == want[:1].tobytes()
        assert len(many.align_db(np.zeros(0, np.float32), np.zeros(1, np.int64))) == 0
        with pytest.raises(S.SfaError, match="single-device"):
            many.align_db_device(0x1000, q_off, n, 0x2000)
    # against the checker as well (not only self-consistency)
    oref = oracle.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    m = min(n, 24)
    assert got[:m].tobytes() == oracle.align_batch(q, q_off[:m + 1], oref, flag, threads=4).tobytes()


def test_sharded_batch_with_long_queries(oracle):
    """Queries beyond 2048 events (row strips, on each shard's second stream beside its wave kernels) mixed with ordinary ones,
    sharded over three contexts on one GPU: rows equal to one device's and to the oracle's."""
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=0)
    rng = np.random.default_rng(21)
    qlens = np.array([2500, 250, 3000, 2049, 100, 4500, 250, 2560, 0, 5000, 250])
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = np.empty(int(q_off[-1]), np.float32)
    for i, l in enumerate(qlens):
        if l == 0:
            continue
        src = ref.forward[0] if i % 2 == 0 else ref.reverse[0]
        st = int(rng.integers(0, len(src) - l))
        seg = src[st:st + l] + rng.normal(scale=0.3, size=l).astype(np.float32)
        q[q_off[i]:q_off[i + 1]] = ((seg - seg.mean()) / seg.std()).astype(np.float32)
    oref = oracle.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    want = oracle.align_batch(q, q_off, oref, flag, threads=8)
    for devices in device_lists()[1:]:
        with S.Aligner(ref, flag, device=0) as one, S.Aligner(ref, flag, devices=devices) as many:
            a = one.align_db(q, q_off)
            b = many.align_db(q, q_off)
            again = many.align_db(q, q_off)
        assert a.tobytes() == b.tobytes() == again.tobytes(), devices
    v = want["valid"] == 1
    assert np.array_equal(a["valid"], want["valid"]) and a[v].tobytes() == want[v].tobytes()


def test_sharded_align_events_and_raw():
    from tests.util import load_case
    from tests.test_host_stages import _pipeline
    c = load_case("dna_default")
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    tables, qs, qe = [], [], []
    for rid, nraw, ev, keep, a, b in _pipeline(c):
        tables.append(ev if keep else None)
        qs.append(a)
        qe.append(b)
    recs = list(S.Blow5File(c["blow5"]))
    raws = [sig for _, _, sig in recs]
    meta = [(m["digitisation"], m["offset"], m["range"]) for _, m, _ in recs]
    raw = np.concatenate(raws)
    off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
    for devices in [[0, 0]] + ([distinct_device_list()] if distinct_device_list() else []):
        with S.Aligner(ref, c["flag"], device=0) as one, S.Aligner(ref, c["flag"], devices=devices) as many:
            assert many.align_events(tables, qs, qe).tobytes() == one.align_events(tables, qs, qe).tobytes()
            r1, i1, e1 = one.align_raw(raw, off, np.array(meta), 50, 250, return_events=True)
            r2, i2, e2 = many.align_raw(raw, off, np.array(meta), 50, 250, return_events=True)
            assert r1.tobytes() == r2.tobytes() and i1.tobytes() == i2.tobytes() and e1.tobytes() == e2.tobytes()


def test_bad_device_list():
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=4, seed=0)
    with pytest.raises(S.SfaError, match="out of range"):
        S.Aligner(ref, flag, devices=[0, 99])
    with pytest.raises(S.SfaError, match="empty device list"):
        S.Aligner(ref, flag, devices=[])


def test_every_device_of_the_box_gives_the_same_rows(oracle):
    """Only where the box has several GPUs: a single-device context on EACH of them (the reference model uploaded to each on
    its own) and the group context over all of them (one upload + peer copies) agree row for row, with each other and with
    the oracle; a second group context in the same process and many small batches through the persistent shard threads too."""
    devs = distinct_device_list()
    if devs is None:
        pytest.skip("one GPU visible: nothing to compare across devices")
    ref, flag, q, q_off, _ = synth.workload("ncov_r9_dna_q250", n_reads=64 * len(devs) + 5, seed=9)
    oref = oracle.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    want = oracle.align_batch(q, q_off, oref, flag, threads=8)
    for d in devs:
        with S.Aligner(ref, flag, device=d) as al:
            assert al.align_db(q, q_off).tobytes() == want.tobytes(), d
    with S.Aligner(ref, flag, devices=devs) as many, S.Aligner(ref, flag, devices=devs[::-1]) as other:
        assert many.n_devices() == len(devs)
        for _ in range(20):  # the shard threads live across calls
            assert many.align_db(q, q_off).tobytes() == want.tobytes()
        many.submit(q, q_off)
        other.submit(q, q_off)  # two group contexts in flight at once
        assert many.wait().tobytes() == want.tobytes() and other.wait().tobytes() == want.tobytes()


def test_shard_threads_survive_many_calls_and_a_failing_shard_reports_its_own_message():
    """One host thread per shard lives as long as the group context (round 2 started a thread per shard and call).  The error
    of a shard's thread reaches the caller (sfa_last_error is per thread): a query window outside the events of a read in the
    LAST shard's range; the context and its threads are fine afterwards."""
    ref, flag, q, q_off, _ = synth.workload("ncov_r9_dna_q250", n_reads=40, seed=2)
    ev = np.zeros(300, dtype=S.EVENT_DTYPE)
    ev["mean"] = np.random.default_rng(1).normal(size=300).astype(np.float32)
    for devices in [[0, 0, 0]] + ([distinct_device_list()] if distinct_device_list() else []):
        with S.Aligner(ref, flag, device=0) as one, S.Aligner(ref, flag, devices=devices) as many:
            want = one.align_db(q, q_off)
            for _ in range(50):
                assert many.align_db(q, q_off).tobytes() == want.tobytes()
            with pytest.raises(S.SfaError, match="outside its 300 events"):
                many.align_events([ev] * 9, [0] * 9, [250] * 8 + [301])
            assert len(many.align_events([ev] * 9, [0] * 9, [250] * 9)) == 9
            assert many.align_db(q, q_off).tobytes() == want.tobytes()
