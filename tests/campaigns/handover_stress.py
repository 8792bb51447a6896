#!/usr/bin/env python3
"""handover_stress.py -- the in-launch hand-overs under LOAD (run on an MI355X box).

fuzz_gpu.py's batches are a few dozen reads: every wave is resident at once and the chip is nearly idle, which is where a
publish / acquire mistake hides (a stale line needs a busy memory system and a consumer on another XCD).  Here the launches
are full-size and ragged: thousands of wave-tasks, pass-2 tickets claimed while fill tasks still drain.  Each iteration runs
the SAME batch through the fused launch (pass 2 inside, hand-over through write-through stores + counters) and through the
separate launches (kernel boundaries in between: no in-launch hand-over at all) and compares the rows byte for byte; a slice
of every batch is also checked against the oracle.  `fused32` does the same for the 32-row fill (queries of 257 .. 2048 events:
snapshots stored write-through to HBM, pass 2 by ticket with its query rows in LDS); `strips` runs the pipelined row strips (one
wave per strip following the strip above) twice, with default and with dense checkpoints and no head start (pass 2 backs off
through them), against each other and the oracle.

Usage: python tests/campaigns/handover_stress.py [iterations] [seed] [fused|fused32|strips]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigfish_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from sigfish_amd import synth  # noqa: E402


def ragged_batch(rng, pool_q, pool_off, n, qmax):
    """n reads drawn from the pool, each cut to a random length (half of them full length): uneven tasks"""
    idx = rng.integers(0, len(pool_off) - 1, size=n)
    lens = np.where(rng.integers(0, 2, n) == 0, qmax, rng.integers(1, qmax + 1, n)).astype(np.int64)
    lens = np.minimum(lens, pool_off[idx + 1] - pool_off[idx])
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q = np.empty(int(q_off[-1]), np.float32)
    for k, i in enumerate(idx):
        q[q_off[k]:q_off[k + 1]] = pool_q[pool_off[i]:pool_off[i] + lens[k]]
    return q, q_off


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    what = sys.argv[3] if len(sys.argv) > 3 else "fused"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    if what == "fused":
        ref, flag, pq, poff, _ = synth.workload("ncov_r9_dna_q250", n_reads=4096, seed=seed)
        oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
        with S.Aligner(ref, flag) as fused, S.Aligner(ref, flag) as plain:
            plain.set_option("fused_trace", 0)
            for it in range(iters):
                n = int(rng.integers(9000, 40000))
                q, q_off = ragged_batch(rng, pq, poff, n, 250)
                margin = int(rng.choice([-1, -1, 0, 40]))  # 0 / 40: pass 2 backs off to the sparse store and the strand start
                for al in (fused, plain):
                    al.set_option("lds_ckpt", 2)
                    al.set_option("trace_margin", margin)
                fused.set_option("fused_trace", 2)
                a = fused.align_db(q, q_off)
                b = plain.align_db(q, q_off)
                ok = a.tobytes() == b.tobytes()
                m = 48
                want = O.align_batch(q, q_off[:m + 1], oref, flag, threads=16)
                ok = ok and a[:m].tobytes() == want.tobytes()
                if not ok:
                    bad += 1
                    diff = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
                    print(f"MISMATCH it={it} n={n} margin={margin} rows={diff[:8]} ({len(diff)} differ)", flush=True)
                if (it + 1) % 10 == 0:
                    print(f"  {it + 1} iterations, {bad} mismatching batches, {time.time() - t0:.0f} s", flush=True)
    elif what == "fused32":
        pools = {}
        for qmax in (500, 1000, 2000):
            ref, flag, pq, poff, _ = synth.workload(f"ncov_r9_dna_q{qmax}", n_reads=1024, seed=seed)
            pools[qmax] = (pq, poff)
        oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
        with S.Aligner(ref, flag) as fused, S.Aligner(ref, flag) as plain:
            plain.set_option("fused_trace", 0)
            fused.set_option("fused_trace", 2)
            for it in range(iters):
                qmax = int(rng.choice([500, 1000, 2000]))
                pq, poff = pools[qmax]
                # more wave-tasks than the chip holds at once (4 096 waves): 4 / 2 / 1 reads per wave, two strands
                n = int(rng.integers(3000, 9000)) * (4 if qmax == 500 else (2 if qmax == 1000 else 1))
                q, q_off = ragged_batch(rng, pq, poff, n, qmax)
                margin = int(rng.choice([-1, -1, -1, 0, 40]))  # 0 / 40: pass 2 backs off through the snapshots
                for al in (fused, plain):
                    al.set_option("trace_margin", margin)
                a = fused.align_db(q, q_off)
                assert fused.profile()["fused_trace"] == 1  # pass 2 ran inside the fill launch
                b = plain.align_db(q, q_off)
                assert plain.profile()["fused_trace"] == 0
                ok = a.tobytes() == b.tobytes()
                m = 24
                want = O.align_batch(q, q_off[:m + 1], oref, flag, threads=16)
                ok = ok and a[:m].tobytes() == want.tobytes()
                if not ok:
                    bad += 1
                    diff = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
                    print(f"MISMATCH it={it} qmax={qmax} n={n} margin={margin} rows={diff[:8]} ({len(diff)} differ)", flush=True)
                if (it + 1) % 5 == 0:
                    print(f"  {it + 1} iterations, {bad} mismatching batches, {time.time() - t0:.0f} s", flush=True)
    else:
        ref, flag, pq, poff, _ = synth.workload("ncov_r9_dna_q4000", n_reads=64, seed=seed)
        oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
        with S.Aligner(ref, flag) as pipe, S.Aligner(ref, flag) as dense:
            for k, v in (("ckpt_interval", 64), ("trace_margin", 0)):  # same strips, every strip of pass 2 backs off through its checkpoints
                dense.set_option(k, v)
            for it in range(iters):
                n = int(rng.integers(300, 1500))
                q, q_off = ragged_batch(rng, pq, poff, n, 4000)
                a = pipe.align_db(q, q_off)
                b = dense.align_db(q, q_off)
                ok = a.tobytes() == b.tobytes()
                if it % 10 == 0:  # (a 4000-event matrix per thread: keep the oracle's share small)
                    want = O.align_batch(q, q_off[:5], oref, flag, threads=4)
                    ok = ok and a[:4].tobytes() == want.tobytes()
                if not ok:
                    bad += 1
                    diff = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
                    print(f"MISMATCH it={it} n={n} rows={diff[:8]} ({len(diff)} differ)", flush=True)
                if (it + 1) % 5 == 0:
                    print(f"  {it + 1} iterations, {bad} mismatching batches, {time.time() - t0:.0f} s", flush=True)
    print(f"{what}: {iters} iterations, {bad} mismatching batches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
