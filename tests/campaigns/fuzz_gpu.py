#!/usr/bin/env python3
"""fuzz_gpu.py -- randomised parity campaign of the HIP path against the oracle (run on an MI355X box):
random reference shapes, ragged query lengths, DNA / RNA / std-DTW / invert, quantised values (exact ties),
random checkpoint intervals, trace margins and lane shapes.  Usage: python tests/campaigns/fuzz_gpu.py [iterations] [seed] [long]
(`long`: query lengths up to 9000 events, i.e. the row-strip path mixed with the wave kernels)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sigfish_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    long_mode = len(sys.argv) > 3 and sys.argv[3] == "long"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    for it in range(iters):
        mode = rng.choice(["dna", "rna", "rna_std", "rna_inv"])
        rna = mode != "dna"
        flag = {"dna": 0, "rna": S.RNA, "rna_std": S.RNA | S.DTW, "rna_inv": S.RNA | S.INV}[mode]
        quant = bool(rng.integers(0, 2))
        nref = int(rng.integers(1, 13))
        lens = [int(x) for x in rng.choice([1, 2, 3, 5, 17, 64, 250, 251, 500, 999, 1024, 1500, 3000, 5000], size=nref)]
        if rng.integers(0, 12) == 0:  # a strand long enough for the sparse HBM checkpoints behind the LDS ones (every 32768 steps)
            lens[int(rng.integers(0, nref))] = int(rng.choice([33000, 40000, 70000]))

        def arr(n):
            return (rng.integers(-8, 9, n) / 4).astype(np.float32) if quant else rng.normal(size=n).astype(np.float32)
        fw = [arr(n) for n in lens]
        rv = None if rna else [arr(n) for n in lens]
        ref = S.RefModel([f"c{i}" for i in range(nref)], [n + 5 for n in lens], lens, rng.integers(0, 4, nref) if rna else [0] * nref, fw, rv)
        n = int(rng.integers(1, 70))
        qmax = int(rng.choice([30, 64, 128, 250, 256, 512, 512, 700, 1024, 1100, 2048]))
        if long_mode:
            qmax = int(rng.choice([2049, 2100, 3000, 4096, 4097, 4100, 6200, 9000]))
        if qmax > 512:
            n = min(n, 24 if qmax <= 2048 else 10)  # keeps the oracle's matrices (one per thread) small
        qlens = rng.integers(0, qmax + 1, size=n)
        if long_mode:
            qlens[0] = qmax
        if rng.integers(0, 3) == 0:
            qlens[:] = qmax
        q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
        q = arr(int(q_off[-1]))
        opts = {}
        if rng.integers(0, 2):
            opts["ckpt_interval"] = int(rng.choice([4, 8, 16, 64, 256, 1024]))
        if rng.integers(0, 2):
            opts["trace_margin"] = int(rng.choice([0, 3, 50, 300]))
        if rng.integers(0, 3) == 0:
            opts["lds_ckpt"] = 0  # every snapshot to HBM (default: rolling in LDS where the shapes allow)
        opts["fused_trace"] = int(rng.choice([0, 1, 2, 2]))  # pass 2 as its own launch / by batch size / inside the fill launch
        if rng.integers(0, 4) == 0:
            opts["lds_ckpt"] = 2  # LDS checkpoints whatever the batch size
        if rng.integers(0, 3) == 0:
            opts["prio_unit"] = int(rng.choice([0, 64, 700]))
        opts["lane_widening"] = int(rng.choice([0, 1, 1, 2, 4]))  # small batches widen by themselves; pin the other shapes too
        if rng.integers(0, 2):  # column segments (small batches): forced counts and short warm-ups exercise the hand-over check
            opts["column_segments"] = int(rng.choice([1, 2, 3, 8, 16]))
            opts["segment_warm_windows"] = int(rng.choice([0, 1, 2, 4]))
        with S.Aligner(ref, flag) as al:
            for k, v in opts.items():
                al.set_option(k, v)
            got = al.align_db(q, q_off)
        oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
        want = O.align_batch(q, q_off, oref, flag, threads=8)
        v = want["valid"] == 1
        ok = np.array_equal(got["valid"], want["valid"]) and got[v].tobytes() == want[v].tobytes()
        if not ok:
            bad += 1
            idx = [i for i in np.nonzero(v)[0] if got[i].tobytes() != want[i].tobytes()][:3]
            print(f"MISMATCH it={it} mode={mode} quant={quant} lens={lens} n={n} opts={opts} reads={idx}")
            for i in idx:
                print("   qlen", qlens[i], "got", got[i], "want", want[i])
        if (it + 1) % 1000 == 0:  # a silent GPU job is taken to be hung
            print(f"  {it + 1} iterations, {bad} mismatching batches so far, {time.time() - t0:.0f} s", flush=True)
    print(f"{iters} iterations, {bad} mismatching batches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
