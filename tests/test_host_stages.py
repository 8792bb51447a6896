"""Host pre-DP stages of the product (BLOW5 reader, event detection, query window + normalisation, k-mer model
reader) against what the COMPILED REFERENCE produced for the same fixture files (tests/golden/cases/*.npz).
CPU only."""
import os

import numpy as np
import pytest

import sigfish_amd as S
from tests.util import GOLD, case_names, load_case


def _pipeline(c):
    rna = bool(c["flag"] & S.RNA)
    f = S.Blow5File(c["blow5"])
    out = []
    for rid, meta, raw in f:
        ev = S.detect_events(raw, meta, rna)
        keep, qs, qe = (False, 0, 0)
        if len(ev):
            keep, qs, qe = S.select_query(ev, raw, meta, c["prefix_size"], c["query_size"], c["flag"], 0)
        out.append((rid, len(raw), ev, keep, qs, qe))
    return out


def test_blow5_header_attrs():
    f = S.Blow5File(os.path.join(GOLD, "data", "sp1_dna.blow5"))
    assert f.attr("experiment_type") == "genomic_dna" and f.attr("sequencing_kit") == "sqk-lsk109"
    assert f.attr("no_such_attr") is None
    g = S.Blow5File(os.path.join(GOLD, "data", "sequin_rna.blow5"))
    assert g.attr("experiment_type") == "rna"
    with pytest.raises(S.SfaError):
        S.Blow5File(os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"))


@pytest.mark.parametrize("name", case_names())
def test_events_and_query_match_reference(name):
    c = load_case(name)
    got = _pipeline(c)
    assert [g[0] for g in got] == [str(x) for x in c["read_ids"]]
    assert [g[1] for g in got] == list(c["len_raw"])
    assert [len(g[2]) for g in got] == list(c["n_events"])
    vi = 0
    for i, (rid, nraw, ev, keep, qs, qe) in enumerate(got):
        assert keep == bool(c["read_valid"][i])
        if not keep:
            continue
        assert (qs, qe) == (int(c["qstart"][i]), int(c["qend"][i])), rid
        q = ev["mean"][qs:qe]
        want = c["queries"][c["q_off"][vi]:c["q_off"][vi + 1]]
        assert np.array_equal(q.view(np.uint32), want.view(np.uint32)), rid   # bit-identical query events
        assert int(ev["start"][qs]) == int(c["ev_start_first"][vi])
        assert int(ev["start"][qe - 1]) == int(c["ev_start_last"][vi])
        assert ev["length"][qe - 1] == c["ev_len_last"][vi]
        vi += 1


def test_kmer_model_text_roundtrip(tmp_path):
    import itertools
    for k in (5, 6):
        lv = np.fromfile(os.path.join(GOLD, "models", f"syn{k}.f32"), np.float32)
        p = tmp_path / f"syn{k}.model"
        with open(p, "w") as f:
            f.write(f"#k\t{k}\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for kmer, v in zip(itertools.product("ACGT", repeat=k), lv):
                f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
        got, kk = S.read_kmer_model(p)
        assert kk == k and np.array_equal(got.view(np.uint32), lv.view(np.uint32))
    bad = tmp_path / "short.model"
    bad.write_text("#k\t5\nAAAAA\t1.0\t1.0\n")
    with pytest.raises(S.SfaError, match="prematurely"):
        S.read_kmer_model(bad)


@pytest.mark.parametrize("name", ["dna_sam", "rna_sam"])
def test_sam_rows_from_reference_alignments(name):
    """SAM writer + host-side warp-path recovery: fed with the reference's own aln_t rows, it must print the
    reference's SAM records (path_to_map / ss string / si tag) byte for byte."""
    c = load_case(name)
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    rows = np.zeros(len(c["rid"]), S.RESULT_DTYPE)
    for f in ("rid", "pos_st", "pos_end", "score", "score2", "strand", "mapq"):
        rows[f] = c[f]
    rows["valid"] = 1
    want = [l + "\n" for l in c["out_text"].split("\n") if l and not l.startswith("@")]
    got, vi = [], 0
    for rid, nraw, ev, keep, qs, qe in _pipeline(c):
        if not keep:
            continue
        r = rows[vi]
        arr = ref.forward[int(r["rid"])] if r["strand"] == ord("+") else ref.reverse[int(r["rid"])]
        got.append(S.sam_row(r, rid, ref.names[int(r["rid"])], ev, qs, qe, arr, int(ref.st_offset[int(r["rid"])]), c["flag"]))
        vi += 1
    assert got == want


@pytest.mark.parametrize("name", ["dna_sam", "rna_sam"])
def test_r2qevent_map_equals_path_to_map_on_the_full_matrix(name, oracle):
    """sfa_r2qevent_map (band re-fill between the winner's start and end columns) against path_to_map (src/sigfish.c:530-571)
    applied to the traceback of the FULL cost matrix, as update_aln does (src/sigfish.c:599-613)."""
    c = load_case(name)
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    rows = np.zeros(len(c["rid"]), S.RESULT_DTYPE)
    for f in ("rid", "pos_st", "pos_end", "score", "score2", "strand", "mapq"):
        rows[f] = c[f]
    rows["valid"] = 1
    vi = 0
    rna = bool(c["flag"] & S.RNA)
    for rid, nraw, ev, keep, qs, qe in _pipeline(c):
        if not keep:
            continue
        r = rows[vi]
        vi += 1
        plus = r["strand"] == ord("+")
        j = int(r["rid"])
        arr = ref.forward[j] if plus else ref.reverse[j]
        got = S.r2qevent_map(r, ev, qs, qe, arr, int(ref.st_offset[j]), c["flag"])
        x = ev["mean"][qs:qe].astype(np.float32)
        if rna:
            x = x[::-1].copy()  # src/sigfish.c:860-866
        off, rl = int(ref.st_offset[j]), len(arr)
        end = int(r["pos_end"]) - off if plus else rl - (int(r["pos_st"]) - off)
        cost = oracle.subsequence(x, arr)
        px, py = oracle.subsequence_path(cost, end)
        n = int(r["pos_end"]) - int(r["pos_st"]) + 1
        assert py[-1] - py[0] + 1 == n and got.shape == (n, 2)
        want = np.full((n, 2), -1, np.int32)
        prev = -1
        for qi, ri in zip(px, py - py[0]):
            if want[ri, 0] == -1:
                want[ri, 0] = qi
            want[ri, 1] = qi
            if prev == qi:
                want[ri] = -1
            prev = qi
        assert np.array_equal(got, want)
    assert vi == len(rows)
    # size query, short buffer, unaligned row
    L = __import__("sigfish_amd._lib", fromlist=["load"]).load()
    bad = rows[0].copy()
    bad["rid"] = -1
    with pytest.raises(S.SfaError):
        S.r2qevent_map(bad, ev, qs, qe, arr, 0, c["flag"])


def test_blow5_reader_rejects_corrupt_files(tmp_path):
    """Flipped bytes, truncation, garbage size fields: an error (or, where the damage hits nothing that is checked, a
    normal read), never a crash or an exception across the C boundary."""
    src = open(os.path.join(GOLD, "data", "sp1_dna.blow5"), "rb").read()
    rng = np.random.default_rng(5)
    path = str(tmp_path / "fz.blow5")
    rejected = 0
    for it in range(300):
        b = bytearray(src)
        if it % 3 == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif it % 3 == 1:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            p = int(rng.integers(64, len(b) - 8))
            b[p:p + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        open(path, "wb").write(b)
        try:
            sum(1 for _ in S.Blow5File(path))
        except Exception:
            rejected += 1
    assert rejected > 250
