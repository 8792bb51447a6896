#!/usr/bin/env python3
"""parity_at_scale.py -- the HIP path against THE REFERENCE ITSELF (oracle/_ref/ref_bench: the reference's own align_db
compiled from its sources) on thousands of synthetic reads per BASELINE workload shape.  Run on an MI355X box:
    python tests/campaigns/parity_at_scale.py [scale [workload:reads ...]]     (scale 1.0 ~ 5 minutes of host CPU on 16 cores)
TEST TOOLING: it loads oracle/ and is not part of the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigfish_amd as S  # noqa: E402
from sigfish_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    cores = len(os.sched_getaffinity(0))
    plan = [("ncov_r9_dna_q250", 20000), ("sequin_r9_rna_q250", 20000), ("rna004_fullref_dtwstd_q250", 5000),
            ("r10_dna_1mb_q250", 96), ("ncov_r9_dna_q1000", 3000), ("ncov_r9_dna_q2000", 1500),
            ("ncov_r9_dna_q3000", 600), ("ncov_r9_dna_q5000", 300)]  # the last two: row strips (queries beyond 2048 events)
    if len(sys.argv) > 2:  # workload:reads pairs instead of the whole plan
        plan = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[2:]]
    bad_total = 0
    for wl, n in plan:
        n = max(8, int(n * scale))
        ref, flag, q, q_off, meta = synth.workload(wl, n_reads=n, seed=4242)
        with S.Aligner(ref, flag) as al:
            t0 = time.time()
            got = al.align_db(q, q_off)
            tg = time.time() - t0
        oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
        res = O.reference_align_batch(q, q_off, oref, flag, threads=cores)
        if res is None:
            print("oracle/_ref/ref_bench is missing: build it where /root/reference exists (make -C oracle ref)")
            sys.exit(2)
        want, secs = res
        v = want["valid"] == 1
        same = np.array_equal(got["valid"], want["valid"]) and got[v].tobytes() == want[v].tobytes()
        nbad = 0 if same else int(sum(got[i].tobytes() != want[i].tobytes() for i in np.nonzero(v)[0]))
        bad_total += nbad
        print(f"{wl:28s} {n:6d} reads  GPU {tg * 1e3:8.1f} ms  reference ({cores} threads) {secs:7.1f} s  "
              f"mapq>0: {int((got['mapq'][v] > 0).sum()):6d}  rows differing from the reference: {nbad}", flush=True)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
