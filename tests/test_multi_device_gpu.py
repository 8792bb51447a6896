"""Multi-GPU behind the C-ABI: sfa_init_devices shards every batch over the listed devices in
contiguous read ranges and returns the rows in input order.  The test box has one GPU, so the device is listed two and
three times (several shards on one GPU, each with its own stream and scratch): the rows must be identical to the
single-device context's, through every host-buffer entry point."""
import numpy as np
import pytest

import sigfish_amd as S
from sigfish_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wl,n", [("ncov_r9_dna_q250", 301), ("sequin_r9_rna_q250", 97), ("rna004_fullref_dtwstd_q250", 23)])
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
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
    with S.Aligner(ref, flag, device=0) as one, S.Aligner(ref, flag, devices=[0, 0, 0]) as many:
        a = one.align_db(q, q_off)
        b = many.align_db(q, q_off)
        again = many.align_db(q, q_off)
    assert a.tobytes() == b.tobytes() == again.tobytes()
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
    with S.Aligner(ref, c["flag"], device=0) as one, S.Aligner(ref, c["flag"], devices=[0, 0]) as many:
        assert many.align_events(tables, qs, qe).tobytes() == one.align_events(tables, qs, qe).tobytes()
        recs = list(S.Blow5File(c["blow5"]))
        raws = [sig for _, _, sig in recs]
        meta = [(m["digitisation"], m["offset"], m["range"]) for _, m, _ in recs]
        raw = np.concatenate(raws)
        off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
        r1, i1, e1 = one.align_raw(raw, off, np.array(meta), 50, 250, return_events=True)
        r2, i2, e2 = many.align_raw(raw, off, np.array(meta), 50, 250, return_events=True)
        assert r1.tobytes() == r2.tobytes() and i1.tobytes() == i2.tobytes() and e1.tobytes() == e2.tobytes()


def test_bad_device_list():
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=4, seed=0)
    with pytest.raises(S.SfaError, match="out of range"):
        S.Aligner(ref, flag, devices=[0, 99])
    with pytest.raises(S.SfaError, match="empty device list"):
        S.Aligner(ref, flag, devices=[])
