"""CPU-side checks of the product library: it loads, exports every symbol include/sigfish_amd.h declares, and
its host helpers (reference event model, z-normalisation, PAF row) agree with the oracle / golden fixtures.
No compute entry point is called here (there is no GPU in this container)."""
import hashlib
import os
import re

import numpy as np
import pytest

import sigfish_amd as S
from sigfish_amd import _lib
from tests.util import ROOT, case_names, load_case


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "sigfish_amd.h")).read()
    declared = set(re.findall(r"\b(sfa_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"sfa_ctx"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert getattr(L, s) is not None
    assert S.version() == "0.1.0"


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ref = S.RefModel(["c"], [10], [5], [0], [np.zeros(5, np.float32)], [np.zeros(5, np.float32)])
    with pytest.raises(S.SfaError, match="no HIP device|CPU fallback"):
        S.Aligner(ref, 0)


def _sha(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, "<f4").tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", case_names())
def test_ref_model_matches_reference_gen_ref(name):
    c = load_case(name)
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    assert ref.names == [str(x) for x in c["ref_names"]]
    assert np.array_equal(ref.ref_lengths, c["ref_lengths"])
    assert np.array_equal(ref.st_offset, c["ref_st_offset"])
    assert _sha(ref.forward) == str(c["fwd_sha256"])
    assert (_sha(ref.reverse) if ref.reverse is not None else "") == str(c["rev_sha256"])


def test_non_acgt_and_lowercase(oracle):
    lv = np.arange(4 ** 5, dtype=np.float32)
    seq = "ACGTNNacgtRYACGTTTGACCAnnACGGT" * 3
    a = S.RefModel.from_records([("x", seq)], lv, 5, 0, 250)
    b = oracle.gen_ref([("x", seq)], lv, 5, 0, 250)
    assert np.array_equal(a.forward[0].view(np.uint32), b.forward[0].view(np.uint32))
    assert np.array_equal(a.reverse[0].view(np.uint32), b.reverse[0].view(np.uint32))


def test_znormalise_matches_oracle(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 25, 250, 29898):
        v = rng.normal(90, 12, n).astype(np.float32)
        assert np.array_equal(S.znormalise(v).view(np.uint32), oracle.normalise(v).view(np.uint32), equal_nan=True)


@pytest.mark.parametrize("name", [n for n in case_names() if "sam" not in n])
def test_paf_rows_from_golden_rows(name):
    """The product's PAF writer reproduces the reference's text from the reference's own aln_t rows."""
    c = load_case(name)
    rows = np.zeros(len(c["rid"]), S.RESULT_DTYPE)
    for f in ("rid", "pos_st", "pos_end", "score", "score2", "strand", "mapq"):
        rows[f] = c[f]
    rows["valid"] = 1
    lines, vi = [], 0
    for i, rid in enumerate(c["read_ids"]):
        if not c["read_valid"][i]:
            continue
        r = rows[vi]
        end_raw = int(c["ev_start_last"][vi]) + int(c["ev_len_last"][vi])
        lines.append(S.paf_row(r, rid, c["ref_names"][int(r["rid"])], int(c["ev_start_first"][vi]), end_raw,
                               int(c["qend"][i]) - 1 - int(c["qstart"][i]), int(c["len_raw"][i]),
                               int(c["ref_seq_lengths"][int(r["rid"])])))
        vi += 1
    assert "".join(lines) == c["out_text"]
