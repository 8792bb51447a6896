#!/usr/bin/env python3
"""The pre-DP stages on the device, ALONE (event detection + query normalisation in front of the alignment):
sfa_align_raw on batches of the reference's DNA fixture reads replicated, one context, nothing else on the GPU.  Prints the
stage timers of the call; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table (tools/raw_stage_profile.sh).
    python tools/raw_stage_profile.py [reads per batch ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sigfish_amd as S  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 32768]
    reads = list(S.Blow5File(os.path.join(GOLD, "data", "sp1_dna.blow5")))
    lv = np.fromfile(os.path.join(GOLD, "models", "syn6.f32"), np.float32)
    ref = S.RefModel.from_fasta(os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), lv, 6, 0, 250)
    with S.Aligner(ref, 0, device=0) as al:
        for n in sizes:
            pick = [reads[i % len(reads)] for i in range(n)]
            raw = np.concatenate([r[2] for r in pick])
            off = np.concatenate([[0], np.cumsum([len(r[2]) for r in pick])]).astype(np.int64)
            sc = np.array([[r[1]["digitisation"], r[1]["offset"], r[1]["range"]] for r in pick], np.float64)
            al.align_raw(raw, off, sc)
            reps = 5
            ev = nm = tot = fill = 0.0
            t0 = time.perf_counter()
            for _ in range(reps):
                al.align_raw(raw, off, sc)
                p = al.profile()
                ev += p["events_ms"]
                nm += p["normalise_ms"]
                tot += p["total_ms"]
                fill += p["fill_ms"]
            wall = (time.perf_counter() - t0) / reps * 1e3
            samples = int(off[-1])
            print(f"{n} reads, {samples} samples ({samples / n:.0f} per read): events {ev / reps:.3f} ms, normalise {nm / reps:.3f} ms, "
                  f"alignment fill {fill / reps:.3f} ms, device total {tot / reps:.3f} ms, call {wall:.3f} ms (host buffers: H2D of "
                  f"{raw.nbytes / 1e6:.0f} MB inside); events stage {samples / (ev / reps * 1e-3) / 1e9:.1f} G samples/s", flush=True)


if __name__ == "__main__":
    main()
