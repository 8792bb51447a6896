#!/usr/bin/env python3
"""Where a short-lived process spends its fixed time (run on the GPU box): import, first / second context, first batches (code
objects are loaded on first use), destroying the contexts, and -- measured by a parent -- leaving the process.
    python tools/init_time.py            the parent: runs the child twice and reports its wall time next to what it printed
    python tools/init_time.py child      the child"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    t0 = time.time()
    import numpy as np  # noqa: F401
    import sigfish_amd as S
    from sigfish_amd import synth
    t1 = time.time()
    ref, flag, q, q_off, meta = synth.workload("ncov_r9_dna_q250", n_reads=4096, seed=7)
    t2 = time.time()
    al = S.Aligner(ref, flag, device=0)
    t3 = time.time()
    al2 = S.Aligner(ref, flag, device=0)
    t4 = time.time()
    al.align_db(q, q_off)
    t5 = time.time()
    al.align_db(q, q_off)
    t6 = time.time()
    al2.align_db(q, q_off)
    t7 = time.time()
    al.close()
    t8 = time.time()
    al2.close()
    t9 = time.time()
    print(f"import {t1 - t0:.3f}  workload {t2 - t1:.3f}  first context {t3 - t2:.3f}  second context {t4 - t3:.3f}  first batch {t5 - t4:.3f}  "
          f"second batch {t6 - t5:.3f}  first batch on ctx2 {t7 - t6:.3f}  destroy ctx1 {t8 - t7:.3f}  destroy ctx2 {t9 - t8:.3f}  "
          f"python total {t9 - t0:.3f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for _ in range(2):
            t0 = time.time()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True)
            dt = time.time() - t0
            print(r.stdout.strip())
            print(f"   process wall {dt:.3f} s (interpreter start-up and exit included)", flush=True)
