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


def test_plan_batch_layout():
    """Host planning logic (the GPU-side replacement of thread.c's fan-out), checked without a GPU."""
    from sigfish_amd.api import plan_batch
    rng = np.random.default_rng(3)
    lens = rng.choice([0, 1, 25, 64, 65, 128, 129, 250, 250, 250, 256, 257, 512], size=1000)
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    info, slot = plan_batch(q_off, [29898, 29898], lane_widening=1)
    assert (slot[lens == 0] == -1).all()
    used = slot[lens > 0]
    assert len(np.unique(used)) == len(used)            # one slot per read
    quads = used >> 2
    rows_per_lane = lambda l: 4 if l <= 64 else 8 if l <= 128 else 16 if l <= 256 else 32
    mixed = 0
    for qd in np.unique(quads):      # a quad holds reads of ONE class whose lengths agree modulo the rows per lane (MixedQuad)
        ls = lens[lens > 0][quads == qd]
        R = rows_per_lane(int(ls.max()))
        assert {rows_per_lane(int(l)) for l in ls} == {R} and len({int(l) % R for l in ls}) == 1, ls
        mixed += len(set(ls)) > 1
    assert mixed >= 1                # 1 and 25 events: both 1 modulo 4
    assert info["n_quads"] == len(np.unique(quads))
    assert info["n_classes"] == 4 and info["max_rows_per_lane"] == 32
    assert 1 <= info["n_chunks"] <= 2 and info["n_tasks"] == info["n_quads"] * info["n_chunks"]
    assert info["ckpt_interval"] == 512 and info["trace_margin"] == 512 + 16
    # long classes come first in task order
    order = np.argsort(used)
    l_sorted = lens[lens > 0][order]
    rpl = np.select([l_sorted <= 64, l_sorted <= 128, l_sorted <= 256], [4, 8, 16], 32)
    assert (np.diff(rpl) <= 0).all()
    # a tight budget raises the interval instead of failing
    info2, _ = plan_batch(q_off, [29898, 29898], ckpt_budget_bytes=1 << 20, lane_widening=1)
    assert info2["ckpt_interval"] > 512 and info2["ckpt_bytes"] <= 1 << 20
    # few reads, many contigs: the contig list is chunked to fill the machine
    info3, _ = plan_batch(np.array([0, 250, 500], np.int64), [375] * 160, lane_widening=1)
    assert info3["n_quads"] == 1 and info3["n_chunks"] == 160
    assert info["max_lanes_per_read"] == 16
    with pytest.raises(S.SfaError):
        plan_batch(np.array([0, 2049], np.int64), [100])  # > SFA_MAX_QUERY


def test_plan_batch_long_queries():
    """Queries of 513..1024 / 1025..2048 events take 32 / 64 lanes: two reads / one read per wave."""
    from sigfish_amd.api import plan_batch
    lens = np.array([2048, 1025, 1025, 1024, 1024, 1024, 513, 512, 512, 250, 250, 250, 250, 250, 600, 600])
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    info, slot = plan_batch(q_off, [5000], lane_widening=1)
    assert info["max_lanes_per_read"] == 64 and info["max_rows_per_lane"] == 32 and info["n_classes"] == 4
    assert info["trace_margin"] == 2048 + 64
    quad, sl = slot >> 2, slot & 3
    assert len(np.unique(slot)) == len(slot)
    assert (sl[lens > 1024] == 0).all()                        # a wave to itself
    assert (sl[(lens > 512) & (lens <= 1024)] <= 1).all()      # two per wave
    for qd in np.unique(quad):
        assert len(set(lens[quad == qd])) == 1
    # waves: 2048 -> 1, 1025 x2 -> 2, 1024 x3 -> 2, 600 x2 -> 1, 513 -> 1, 512 x2 -> 1, 250 x5 -> 2
    assert info["n_quads"] == 10
    # long classes first
    order = np.argsort(slot)
    span = np.select([lens[order] <= 64, lens[order] <= 128, lens[order] <= 256, lens[order] <= 512, lens[order] <= 1024],
                     [64, 128, 256, 512, 1024], 2048)
    assert (np.diff(span) <= 0).all()
    info2, _ = plan_batch(np.array([0, 700], np.int64), [100], lane_widening=1)
    assert info2["max_lanes_per_read"] == 32 and info2["trace_margin"] == 700 + 32


def test_plan_batch_lane_widening():
    """Small batches trade rows per lane for lanes per read: 2-4x the waves, each with a 2-4x shorter step."""
    from sigfish_amd.api import plan_batch
    lens = np.array([250] * 10 + [100] * 3 + [500, 60, 1000, 2000])
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    i1, s1 = plan_batch(q_off, [29898, 29898], lane_widening=1)
    i2, s2 = plan_batch(q_off, [29898, 29898], lane_widening=2)
    i4, s4 = plan_batch(q_off, [29898, 29898], lane_widening=4)
    # w=1: 2000 -> 1 wave, 1000 -> 1, 500 -> 1, 250x10 -> 3, 100x3 -> 1, 60 -> 1
    assert (i1["n_quads"], i1["lane_widening"], i1["max_lanes_per_read"]) == (8, 1, 64)
    # w=2: 250 -> (8 rows, 32 lanes): 5 waves; 100 -> (4,32): 2 waves; 500 -> (16,32); 60 stays (4,16); 1000 -> (16,64)
    assert (i2["n_quads"], i2["lane_widening"]) == (1 + 1 + 1 + 5 + 2 + 1, 2)
    # w=4: 250 -> (4,64): 10 waves; 100 -> (4,32) (rows per lane cannot drop below 4): 2; 500 -> (8,64)
    assert (i4["n_quads"], i4["lane_widening"]) == (1 + 1 + 1 + 10 + 2 + 1, 4)
    for s, per250 in ((s1, 4), (s2, 2), (s4, 1)):
        assert len(np.unique(s)) == len(s)
        assert ((s[:10] & 3) < per250).all()
    # auto (1024 SIMDs assumed; t = wave-tasks per SIMD with the 16-lane shapes = reads / 2048 on two strands): x4 below
    # t = 0.9, x2 below t = 4, the 16-lane shapes from there on
    assert plan_batch(q_off, [29898, 29898])[0]["lane_widening"] == 4
    big = np.arange(0, 250 * 20001, 250, dtype=np.int64)
    assert plan_batch(big, [29898, 29898])[0]["lane_widening"] == 1
    mid = np.arange(0, 250 * 10241, 250, dtype=np.int64)
    assert plan_batch(mid, [29898, 29898])[0]["lane_widening"] == 1
    assert plan_batch(mid[:8193], [29898, 29898])[0]["lane_widening"] == 1   # 8 192 reads: t = 4
    assert plan_batch(mid[:8001], [29898, 29898])[0]["lane_widening"] == 2
    assert plan_batch(mid[:2049], [29898, 29898])[0]["lane_widening"] == 2   # 2 048 reads: t = 1
    assert plan_batch(mid[:1801], [29898, 29898])[0]["lane_widening"] == 4
    with pytest.raises(S.SfaError):
        plan_batch(q_off, [100], lane_widening=3)


def test_plan_batch_empty_and_single():
    from sigfish_amd.api import plan_batch
    info, slot = plan_batch(np.array([0, 0, 0], np.int64), [50])
    assert info["n_quads"] == 0 and (slot == -1).all()
    info, slot = plan_batch(np.array([0, 7], np.int64), [50])
    assert info["n_quads"] == 1 and slot[0] == 0


def test_plan_batch_ragged_lengths_share_waves():
    """A ragged batch (hundreds of distinct lengths, a read or two of each) used to take a wave per length; with reads whose
    lengths agree modulo the rows per lane sharing waves, a class leaves at most that many partly filled waves."""
    from sigfish_amd.api import plan_batch
    rng = np.random.default_rng(5)
    lens = np.concatenate([np.full(7782, 250), rng.integers(25, 250, size=410)])  # 8 192 reads, 5 % shorter
    rng.shuffle(lens)
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    info, slot = plan_batch(q_off, [29898, 29898], lane_widening=1)
    assert len(np.unique(slot)) == len(slot)
    # reads / 4 plus at most one partly filled wave per residue of each class: 4 + 8 + 16
    assert info["n_quads"] <= (len(lens) + 3) // 4 + 28
    assert info["n_quads"] <= 1.03 * len(lens) / 4     # (one wave per length would be ~2 170)
    # inside a wave the longest read comes first (it fixes where the last query row sits)
    order = np.argsort(slot)
    quads = slot[order] >> 2
    for qd in np.unique(quads)[:200]:
        ls = lens[order][quads == qd]
        assert (np.diff(ls) <= 0).all()


def test_plan_batch_invariants_on_random_batches():
    """Property test of the planner (hypothesis): whatever the lengths, every non-empty read gets exactly one slot, empty reads
    none, the reads of a wave belong to one class and agree modulo its rows per lane (MixedQuad's condition), the longest comes
    first, and the task count is quads x chunks."""
    from hypothesis import given, settings, strategies as st
    from sigfish_amd.api import plan_batch

    rows_per_lane = lambda l: 4 if l <= 64 else 8 if l <= 128 else 16 if l <= 256 else 32

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.one_of(st.integers(0, 300), st.integers(0, 2048), st.sampled_from([0, 1, 4, 64, 65, 250, 256, 257, 1024, 2048])),
                    min_size=0, max_size=400),
           st.lists(st.integers(1, 5000), min_size=1, max_size=12))
    def check(lens, ref_lens):
        lens = np.asarray(lens, np.int64)
        q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        info, slot = plan_batch(q_off, ref_lens, lane_widening=1)
        assert (slot[lens == 0] == -1).all()
        used = slot[lens > 0]
        assert (used >= 0).all() and len(np.unique(used)) == len(used)
        if len(used) == 0:
            assert info["n_quads"] == 0
            return
        order = np.argsort(used)
        quads = used[order] >> 2
        ls_sorted = lens[lens > 0][order]
        assert info["n_quads"] == len(np.unique(quads))
        assert info["n_tasks"] == info["n_quads"] * info["n_chunks"]
        for qd in np.unique(quads):
            ls = ls_sorted[quads == qd]
            R = rows_per_lane(int(ls.max()))
            assert {rows_per_lane(int(l)) for l in ls} == {R}
            assert len({int(l) % R for l in ls}) == 1
            assert (np.diff(ls) <= 0).all()

    check()
