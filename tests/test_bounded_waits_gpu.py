"""Waits between waves of one launch are bounded: a hand-over that never comes is SFA_EKERNEL, not a hung stream.

The fused launch (pass 2 waits for its quad's fill tasks) and the pipelined row strips (a strip waits for the row above) spin
on counters other waves bump.  A logic slip there used to be an endless loop on the device; now every spin gives up after
`spin_limit_ms` of wall clock, leaves an error word, and the batch call fails loudly.  The test hooks `debug_drop_quad` /
`debug_drop_strip` make one producer skip its signal; each case is ONE launch with a short limit."""
import numpy as np
import pytest

import sigfish_amd as S
from sigfish_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_ref(O, ref):
    return O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)


def _same(got, want):
    for f in ("rid", "pos_st", "pos_end", "strand", "mapq", "valid"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))
    assert np.array_equal(got["score2"].view(np.uint32), want["score2"].view(np.uint32))


@pytest.fixture(autouse=True)
def _hooks(monkeypatch):
    monkeypatch.setenv("SFA_TEST_HOOKS", "1")  # debug_drop_* are refused otherwise (sfa_set_option reads the environment)


def test_the_hooks_are_refused_without_the_environment_switch(monkeypatch):
    monkeypatch.delenv("SFA_TEST_HOOKS")
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=4, seed=1)
    with S.Aligner(ref, flag) as al:
        for key in ("debug_drop_quad", "debug_drop_strip"):
            with pytest.raises(S.SfaError, match="test hook"):
                al.set_option(key, 0)


@pytest.mark.parametrize("wl", ["ncov_r9_dna_q250", "ncov_r9_dna_q500"])  # the LDS-checkpoint fill / the 32-row fill (snapshots in HBM)
def test_a_quad_that_never_publishes_fails_the_batch_instead_of_hanging(oracle, wl):
    ref, flag, q, q_off, _ = synth.workload(wl, n_reads=64, seed=3)
    with S.Aligner(ref, flag) as al:
        for k, v in (("lane_widening", 1), ("lds_ckpt", 2), ("fused_trace", 2), ("spin_limit_ms", 300), ("debug_drop_quad", 0)):
            al.set_option(k, v)
        with pytest.raises(S.SfaError, match=r"pass 2 of quad 0 waited 300 ms"):
            al.align_db(q, q_off)
        # the context survives: same batch, producer restored
        al.set_option("debug_drop_quad", -1)
        got = al.align_db(q, q_off)
    _same(got, oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), flag, threads=8))


def test_a_strip_that_never_publishes_fails_the_batch_instead_of_hanging(oracle):
    rng = np.random.default_rng(5)
    ref = S.RefModel(["a", "b"], [4005, 3005], [4000, 3000], [0, 0], [rng.normal(size=4000).astype(np.float32), rng.normal(size=3000).astype(np.float32)],
                     [rng.normal(size=4000).astype(np.float32), rng.normal(size=3000).astype(np.float32)])
    qlens = np.array([3000, 100, 2500], np.int64)
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    q = rng.normal(size=int(q_off[-1])).astype(np.float32)
    with S.Aligner(ref, 0) as al:
        al.set_option("spin_limit_ms", 300)
        al.set_option("debug_drop_strip", 0)
        with pytest.raises(S.SfaError, match=r"row strips: a strip waited 300 ms"):
            al.align_db(q, q_off)
        al.set_option("debug_drop_strip", -1)
        got = al.align_db(q, q_off)
    _same(got, oracle.align_batch(q, q_off, _oracle_ref(oracle, ref), 0, threads=8))


def test_spin_limit_is_validated():
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=4, seed=1)
    with S.Aligner(ref, flag) as al:
        with pytest.raises(S.SfaError):
            al.set_option("spin_limit_ms", 0)
