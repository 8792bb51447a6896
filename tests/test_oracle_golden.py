"""The oracle (CPU restatement, oracle/sdtw_oracle.c) against the committed golden fixtures that were produced
by the COMPILED REFERENCE (oracle/make_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import hashlib
import os

import numpy as np
import pytest

from tests.util import GOLD, case_names, load_case, paf_lines_from_results

CASES = case_names()


def _sha(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, "<f4").tobytes())
    return h.hexdigest()


def test_fixture_inventory():
    assert len(CASES) >= 14
    for f in ("nCoV-2019.reference.fasta", "rnasequin_sequences_2.4.fa", "sp1_dna.blow5", "sequin_rna.blow5"):
        assert os.path.exists(os.path.join(GOLD, "data", f))


@pytest.mark.parametrize("name", CASES)
def test_gen_ref_matches_reference_arrays(oracle, name):
    c = load_case(name)
    ref = oracle.gen_ref(oracle.read_fasta(c["fasta"]), c["levels"], c["k"], c["flag"], c["query_size"])
    assert list(ref.names) == [str(x) for x in c["ref_names"]]
    assert np.array_equal(ref.ref_lengths, c["ref_lengths"])
    assert np.array_equal(ref.seq_lengths, c["ref_seq_lengths"])
    assert np.array_equal(ref.st_offset, c["ref_st_offset"])
    assert _sha(ref.forward) == str(c["fwd_sha256"])
    if ref.reverse is not None:
        assert _sha(ref.reverse) == str(c["rev_sha256"])
    else:
        assert str(c["rev_sha256"]) == ""


@pytest.mark.parametrize("name", CASES)
def test_alignment_rows_and_text(oracle, name):
    c = load_case(name)
    ref = oracle.gen_ref(oracle.read_fasta(c["fasta"]), c["levels"], c["k"], c["flag"], c["query_size"])
    res = oracle.align_batch(c["queries"], c["q_off"], ref, c["flag"], threads=4)
    for f in ("rid", "pos_st", "pos_end", "mapq", "strand"):
        assert np.array_equal(res[f], c[f]), f
    # scores are bit-exact, not merely close
    assert np.array_equal(res["score"].view(np.uint32), c["score"].view(np.uint32))
    assert np.array_equal(res["score2"].view(np.uint32), c["score2"].view(np.uint32))
    if not c["sam"]:
        text = paf_lines_from_results(oracle, c, res, ref.names, ref.seq_lengths)
        assert text == c["out_text"]


@pytest.mark.parametrize("name", CASES)
def test_query_window(oracle, name):
    c = load_case(name)
    if c["prefix_size"] < 0:
        pytest.skip("auto prefix needs the adaptor/poly-A segmenter (not on the DP path)")
    for i in range(len(c["read_ids"])):
        keep, qs, qe = oracle.query_window(int(c["n_events"][i]), c["prefix_size"], c["query_size"], c["flag"])
        assert bool(keep) == bool(c["read_valid"][i])
        if keep:
            assert (qs, qe) == (int(c["qstart"][i]), int(c["qend"][i]))


def test_kernel_vectors(oracle):
    z = np.load(os.path.join(GOLD, "kernel_vectors.npz"))
    n = int(z["count"])
    assert n >= 30
    for i in range(n):
        x, y = z[f"x{i}"], z[f"y{i}"]
        cs, cd = oracle.subsequence(x, y), oracle.std_dtw(x, y)
        assert np.array_equal(cs[-1].view(np.uint32), z[f"sub_last{i}"].view(np.uint32))
        assert np.array_equal(cd[-1].view(np.uint32), z[f"std_last{i}"].view(np.uint32))
        m = len(y)
        assert [oracle.path_start(cs, j) for j in range(m)] == list(z[f"sub_start{i}"])
        assert [oracle.path_start(cd, j) for j in range(m)] == list(z[f"std_start{i}"])
        if f"sub_full{i}" in z.files:
            assert np.array_equal(cs.view(np.uint32), z[f"sub_full{i}"].view(np.uint32))
            assert np.array_equal(cd.view(np.uint32), z[f"std_full{i}"].view(np.uint32))
            px, py = oracle.subsequence_path(cs, m - 1)
            assert np.array_equal(px, z[f"sub_px{i}"]) and np.array_equal(py, z[f"sub_py{i}"])


def test_mapq_edges(oracle):
    assert oracle.mapq(10.0, 10.0) == 0
    assert oracle.mapq(10.0, 11.0) == 50
    assert oracle.mapq(10.0, 20.0) == 60
    assert oracle.mapq(10.0, float("inf")) == 0  # (int)round(inf) -> INT_MIN -> u8 0 on x86-64
    assert oracle.mapq(0.0, 0.0) == 0            # NaN
