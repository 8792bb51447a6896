#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Degenerate queries at the align_db boundary, answered by THE COMPILED REFERENCE (oracle/_ref/ref_bench =
its own dtw_single over work_db): what does sigfish do with a constant query, with exact ties, with a NaN or inf event?

Writes tests/golden/degenerate/degenerate.npz (queries, offsets, the reference's rows for the batch of FINITE reads, indices
of the reads that were made non-finite) and reference_on_non_finite.txt (the reference's stderr and exit status for each
non-finite case: it aborts in update_aln, src/sigfish.c:611).  Run in the build container (needs /root/reference built
into oracle/_ref)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from sigfish_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "degenerate")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref, flag, q, q_off, _ = synth.workload("ncov_r9_dna_q250", n_reads=16, seed=2)
    oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    q = q.copy()
    sl = lambda i: slice(int(q_off[i]), int(q_off[i + 1]))  # noqa: E731
    # finite but degenerate: the reference prints rows for these
    q[sl(9)] = 0.0                                   # constant query (a zero-variance window before the division)
    q[sl(10)] = 1.5                                  # constant, non-zero
    q[sl(11)] = np.round(q[sl(11)] * 2) / 2          # quantised: exact ties between windows and inside the traceback
    q[sl(12)] = np.tile(np.float32([-1.0, 1.0]), 125)[: int(q_off[13] - q_off[12])]  # two alternating levels
    # non-finite: one NaN event, all NaN (what (x - mean) / 0 gives for a constant window), one +inf, one -inf
    cases = {0: "one NaN event", 3: "one +inf event", 6: "all events NaN", 14: "one -inf event"}
    bad = q.copy()
    bad[int(q_off[0]) + 5] = np.nan
    bad[int(q_off[3]) + 7] = np.inf
    bad[sl(6)] = np.nan
    bad[int(q_off[14]) + 100] = -np.inf
    log = []
    for i, what in cases.items():
        one = bad[sl(i)]
        try:
            O.reference_align_batch(one, np.array([0, len(one)], np.int64), oref, flag, threads=1)
            log.append(f"read {i} ({what}): the reference returned normally\n")
        except subprocess.CalledProcessError as e:
            log.append(f"read {i} ({what}): the reference exited with status {e.returncode}; stderr:\n{e.stderr.decode()}\n")
    open(os.path.join(OUT, "reference_on_non_finite.txt"), "w").write("".join(log))
    rows, _ = O.reference_align_batch(q, q_off, oref, flag, threads=4)  # the finite batch: all 16 reads
    want = rows.copy()
    for i in cases:  # what the library must return for the batch with the non-finite reads in it: those reads skipped
        want[i] = np.zeros(1, O.RESULT_DTYPE)[0]
        want["rid"][i] = want["pos_st"][i] = want["pos_end"][i] = -1
        want["score"][i] = want["score2"][i] = np.inf
    np.savez_compressed(os.path.join(OUT, "degenerate.npz"), queries_finite=q, queries=bad, q_off=q_off, rows_finite=rows, rows=want,
                        non_finite=np.array(sorted(cases), np.int32))
    print(open(os.path.join(OUT, "reference_on_non_finite.txt")).read())
    print(rows[[9, 10, 11, 12]])


if __name__ == "__main__":
    main()
