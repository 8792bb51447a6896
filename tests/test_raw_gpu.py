"""Raw signal in, rows out (sfa_align_raw): event detection, query window and normalisation on the GPU.  Checked
against the fixtures produced by the compiled reference: event counts, windows, raw coordinates, alignment rows and
the final PAF text."""
import numpy as np
import pytest

import sigfish_amd as S
from tests.util import case_names, load_case

pytestmark = pytest.mark.gpu

CASES = [n for n in case_names() if "sam" not in n and "pauto" not in n]


def _load_raw(path):
    ids, raws, scal = [], [], []
    for rid, meta, raw in S.Blow5File(path):
        ids.append(rid)
        raws.append(raw)
        scal.append([meta["digitisation"], meta["offset"], meta["range"]])
    off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
    return ids, np.concatenate(raws), off, np.array(scal, np.float64)


@pytest.mark.parametrize("name", CASES)
def test_raw_path_matches_reference(name):
    c = load_case(name)
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    ids, raw, off, scal = _load_raw(c["blow5"])
    with S.Aligner(ref, c["flag"]) as al:
        rows, info = al.align_raw(raw, off, scal, c["prefix_size"], c["query_size"])
    assert list(info["n_events"]) == list(c["n_events"])
    assert list(rows["valid"] == 1) == list(c["read_valid"])
    v = c["read_valid"]
    assert np.array_equal(info["qstart"][v], c["qstart"][v]) and np.array_equal(info["qend"][v], c["qend"][v])
    assert np.array_equal(info["start_raw_idx"][v], c["ev_start_first"])
    for f in ("rid", "pos_st", "pos_end", "mapq", "strand"):
        assert np.array_equal(rows[f][v], c[f]), f
    assert np.array_equal(rows["score"][v].view(np.uint32), c["score"].view(np.uint32))
    assert np.array_equal(rows["score2"][v].view(np.uint32), c["score2"].view(np.uint32))
    lines = []
    for i, rid in enumerate(ids):
        if not v[i]:
            continue
        r = rows[i]
        lines.append(S.paf_row(r, rid, ref.names[int(r["rid"])], int(info["start_raw_idx"][i]), int(info["end_raw_idx"][i]),
                               int(info["qend"][i]) - 1 - int(info["qstart"][i]), int(off[i + 1] - off[i]),
                               int(ref.seq_lengths[int(r["rid"])])))
    assert "".join(lines) == c["out_text"]


def test_raw_path_many_reads_equals_host_path():
    """2 000 reads (the DNA fixture replicated, some truncated): device events == host events, row for row."""
    c = load_case("dna_default")
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    ids, raw, off, scal = _load_raw(c["blow5"])
    rng = np.random.default_rng(1)
    raws, scs = [], []
    for k in range(2000):
        i = k % 5
        r = raw[off[i]:off[i + 1]]
        if k % 7 == 3:
            r = r[:int(rng.integers(0, len(r)))]      # truncated, incl. empty and too short reads
        raws.append(r)
        scs.append(scal[i])
    off2 = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
    raw2 = np.concatenate(raws)
    with S.Aligner(ref, 0) as al:
        rows, info = al.align_raw(raw2, off2, np.array(scs), 50, 250)
        # host stages + align_events on the same reads
        tabs, qs, qe = [], [], []
        for k, r in enumerate(raws):
            meta = dict(digitisation=scs[k][0], offset=scs[k][1], range=scs[k][2])
            ev = S.detect_events(r, meta, False) if len(r) else np.zeros(0, S.EVENT_DTYPE)
            keep, a, b = (False, 0, 0)
            if len(ev):
                keep, a, b = S.select_query(ev, r, meta, 50, 250, 0, 0)
            tabs.append(ev if keep else None)
            qs.append(a if keep else 0)
            qe.append(b if keep else 0)
            assert info["n_events"][k] == len(ev)
        want = al.align_events(tabs, qs, qe)
    assert np.array_equal(rows["valid"], want["valid"])
    m = want["valid"] == 1
    assert rows[m].tobytes() == want[m].tobytes()
    assert np.array_equal(info["qstart"][m], np.array(qs)[m]) and np.array_equal(info["qend"][m], np.array(qe)[m])


def test_parallel_prefix_sums_and_their_fallback():
    """The wave-per-read prefix sums run only where no addition can round; reads that fail the certificate (here: a tiny
    offset turns raw == 0 into 1e-30 pA next to ordinary values, > 52 bits apart) go through the sequential kernel.
    Both routes and the all-sequential setting give the same events, windows and rows as the host stages."""
    c = load_case("dna_default")
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    ids, raw, off, scal = _load_raw(c["blow5"])
    raws, scs = [], []
    for k in range(130):  # more than two waves of reads, alternating certified / uncertified
        i = k % 5
        r = raw[off[i]:off[i + 1]].copy()
        sc = np.array(scal[i], np.float64)
        if k % 2 == 1:
            sc[1] = 1e-30     # offset: (float)0 + 1e-30 is a tiny non-zero pA value
            r[::97] = 0
        raws.append(r)
        scs.append(sc)
    off2 = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
    raw2 = np.concatenate(raws)
    with S.Aligner(ref, 0) as al:
        rows_par, info_par = al.align_raw(raw2, off2, np.array(scs), 50, 250)
        al.set_option("ev_parallel", 2)  # sequential prefix sums for every read
        rows_seq, info_seq = al.align_raw(raw2, off2, np.array(scs), 50, 250)
    assert rows_par.tobytes() == rows_seq.tobytes() and info_par.tobytes() == info_seq.tobytes()
    for k in (0, 1, 2, 3, 64, 65, 129):  # host stages on a few of each kind
        meta = dict(digitisation=scs[k][0], offset=scs[k][1], range=scs[k][2])
        ev = S.detect_events(raws[k], meta, False)
        assert info_par["n_events"][k] == len(ev)


def test_query_window_events_come_back_for_sam():
    """sfa_align_raw_ex: the event tables of the query windows (means z-normalised) equal the host stages' tables."""
    c = load_case("dna_default")
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    ids, raw, off, scal = _load_raw(c["blow5"])
    with S.Aligner(ref, 0) as al:
        rows, info, qev = al.align_raw(raw, off, scal, 50, 250, return_events=True)
        rows2, info2 = al.align_raw(raw, off, scal, 50, 250)
    assert rows.tobytes() == rows2.tobytes() and info.tobytes() == info2.tobytes()
    for i in range(len(ids)):
        r = raw[off[i]:off[i + 1]]
        meta = dict(digitisation=scal[i][0], offset=scal[i][1], range=scal[i][2])
        ev = S.detect_events(r, meta, False)
        keep, a, b = S.select_query(ev, r, meta, 50, 250, 0, 0)   # normalises ev[a:b].mean in place
        assert keep and (a, b) == (info["qstart"][i], info["qend"][i])
        got = qev[i][:b - a]
        for f in ("start", "length", "mean", "stdv"):
            assert np.array_equal(got[f], ev[f][a:b]), (i, f)


def test_parallel_peak_picker_and_its_fallback():
    """The chunk-parallel peak picker is accepted only where every lane's walk re-joins the sequential one; reads it
    declines (here: too short or too long to split) go through the sequential kernel.  Either way, and with the option
    off, the same events, windows and rows."""
    c = load_case("dna_default")
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    ids, raw, off, scal = _load_raw(c["blow5"])
    raws, scs = [], []
    for k in range(70):
        i = k % 5
        r = raw[off[i]:off[i + 1]]
        if k % 4 == 1:
            r = r[:1200]                      # 18 samples per lane: below the speculation threshold
        elif k % 4 == 2:
            r = np.tile(r, 5)[:20000]         # more than 288 samples per lane: above it
        raws.append(r)
        scs.append(scal[i])
    off2 = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
    raw2 = np.concatenate(raws)
    with S.Aligner(ref, 0) as al:
        rows_par, info_par, ev_par = al.align_raw(raw2, off2, np.array(scs), 50, 250, return_events=True)
        al.set_option("ev_parallel", 1)  # sequential peak picker for every read
        rows_seq, info_seq, ev_seq = al.align_raw(raw2, off2, np.array(scs), 50, 250, return_events=True)
    assert rows_par.tobytes() == rows_seq.tobytes() and info_par.tobytes() == info_seq.tobytes() and ev_par.tobytes() == ev_seq.tobytes()
    for k in (0, 1, 2, 3, 64, 69):
        meta = dict(digitisation=scs[k][0], offset=scs[k][1], range=scs[k][2])
        assert info_par["n_events"][k] == len(S.detect_events(raws[k], meta, False))
