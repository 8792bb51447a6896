"""The oracle against the COMPILED REFERENCE (oracle/_ref, built from /root/reference where the sources lie).
Only runs in the build container; skipped elsewhere (the GPU box has no reference tree)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import GOLD, case_names, load_case

pytestmark = pytest.mark.skipif(not os.path.exists(O.REF_SO), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("seed", range(40))
def test_cdtw_randomised(seed):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(1, 60)), int(rng.integers(1, 300))
    if seed % 3 == 0:
        x = (rng.integers(-3, 4, n) / 2).astype(np.float32)
        y = (rng.integers(-3, 4, m) / 2).astype(np.float32)
    else:
        x, y = rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32)
    for mine, theirs in ((O.subsequence, O.ref_subsequence), (O.std_dtw, O.ref_std_dtw)):
        a, b = mine(x, y), theirs(x, y)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for j in rng.integers(0, m, 8):
            px, py = O.subsequence_path(a, int(j))
            rx, ry = O.ref_subsequence_path(b, int(j))
            assert np.array_equal(px, rx) and np.array_equal(py, ry)


@pytest.mark.parametrize("name", ["dna_default", "rna_full_ref_dtw_std", "rna_from_end"])
def test_driver_reproduces_golden(name, tmp_path):
    """The committed fixtures are exactly what the reference build prints today."""
    c = load_case(name)
    cmd = [O.REF_DRIVER, "--model", os.path.join(GOLD, "models", f"syn{c['k']}.f32"), "--kmer", str(c["k"]),
           *[str(a) for a in c["args"]], c["fasta"], c["blow5"]]
    out = subprocess.run(cmd, check=True, capture_output=True).stdout.decode()
    assert out == c["out_text"]


@pytest.mark.parametrize("wl,n", [("ncov_r9_dna_q250", 12), ("sequin_r9_rna_q250", 16), ("rna004_fullref_dtwstd_q250", 8)])
def test_reference_align_db_on_synthetic_workloads(wl, n):
    """The reference's own dtw_single/work_db (oracle/_ref/ref_bench) on the synthetic BASELINE workloads: the oracle
    must produce the same rows -- this pins the checker used by the GPU tests on inputs the fixtures do not cover."""
    from sigfish_amd import synth
    ref, flag, q, q_off, meta = synth.workload(wl, n_reads=n, seed=21)
    oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    got = O.reference_align_batch(q, q_off, oref, flag, threads=4)
    assert got is not None
    rows, secs = got
    want = O.align_batch(q, q_off, oref, flag, threads=4)
    assert rows.tobytes() == want.tobytes() and secs > 0
