#!/usr/bin/env python3
"""fuzz_raw_gpu.py -- randomised parity of the device-side pre-DP stages (sfa_align_raw: prefix sums, t-statistics, peak
picker, event statistics, query window, normalisation, then the alignment) against the host stages, which are pinned
bit-exactly to the compiled reference by tests/test_host_stages.py.  Synthetic step signals with random dwell, noise,
length (incl. empty / too short reads), scaling and DNA / RNA detector parameters.
Usage (MI355X box): python tools/fuzz_raw_gpu.py [iterations] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigfish_amd as S  # noqa: E402


def signal(rng, n):
    if n == 0:
        return np.zeros(0, np.int16)
    dwell = rng.integers(2, int(rng.integers(4, 40)), size=n // 2 + 2)
    levels = rng.normal(rng.uniform(300, 700), rng.uniform(20, 120), size=len(dwell))
    x = np.repeat(levels, dwell)[:n]
    if len(x) < n:
        x = np.concatenate([x, np.full(n - len(x), x[-1] if len(x) else 500.0)])
    x = x + rng.normal(0, rng.uniform(0.5, 15), size=n)
    if rng.integers(0, 6) == 0:
        x[:] = np.round(x / 8) * 8  # coarse quantisation: exact ties in the t-statistics
    return np.clip(np.round(x), -2000, 4000).astype(np.int16)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    for it in range(iters):
        rna = bool(rng.integers(0, 2))
        flag = S.RNA if rna else 0
        lens = [int(x) for x in rng.integers(300, 3000, size=int(rng.integers(1, 4)))]
        fw = [rng.normal(size=n).astype(np.float32) for n in lens]
        rv = None if rna else [rng.normal(size=n).astype(np.float32) for n in lens]
        ref = S.RefModel([f"c{i}" for i in range(len(lens))], [n + 5 for n in lens], lens, [0] * len(lens), fw, rv)
        n = int(rng.integers(1, 150))
        raws, scs = [], []
        for _ in range(n):
            ln = int(rng.choice([0, 1, 5, 30, 200, 1000, 3000, 6000, 12000], p=[.03, .02, .02, .03, .1, .2, .3, .2, .1]))
            ln = int(ln * rng.uniform(0.5, 1.5))
            raws.append(signal(rng, ln))
            scs.append([float(rng.choice([2048.0, 8192.0])), float(rng.integers(-20, 40)), float(rng.uniform(700, 1500))])
        prefix, query = int(rng.choice([0, 10, 50])), int(rng.choice([30, 100, 250, 600]))
        off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
        cat = np.concatenate(raws) if off[-1] else np.zeros(0, np.int16)
        with S.Aligner(ref, flag) as al:
            al.set_option("ev_parallel", int(rng.choice([3, 3, 3, 2, 1, 0])))  # wave-per-read prefix sums (bit 0) / peak picker (bit 1), or the sequential kernels
            rows, info, qev = al.align_raw(cat, off, np.array(scs), prefix, query, return_events=True)
            tabs, qs, qe = [], [], []
            for k, r in enumerate(raws):
                meta = dict(digitisation=scs[k][0], offset=scs[k][1], range=scs[k][2])
                ev = S.detect_events(r, meta, rna) if len(r) else np.zeros(0, S.EVENT_DTYPE)
                keep, a, b = (False, 0, 0)
                if len(ev):
                    keep, a, b = S.select_query(ev, r, meta, prefix, query, flag, 0)
                if info["n_events"][k] != len(ev):
                    bad += 1
                    print(f"MISMATCH it={it} read {k} (len {len(r)}): {info['n_events'][k]} events on the device, {len(ev)} on the host")
                    break
                if keep and not all(np.array_equal(qev[k][:b - a][f], ev[f][a:b]) for f in ("start", "length", "mean")):
                    bad += 1
                    print(f"MISMATCH it={it} read {k}: query window events differ")
                    break
                tabs.append(ev if keep else None)
                qs.append(a if keep else 0)
                qe.append(b if keep else 0)
            else:
                want = al.align_events(tabs, qs, qe)
                m = want["valid"] == 1
                if not (np.array_equal(rows["valid"], want["valid"]) and rows[m].tobytes() == want[m].tobytes()):
                    bad += 1
                    print(f"MISMATCH it={it}: rows differ (rna={rna}, n={n}, prefix={prefix}, query={query})")
        if (it + 1) % 10 == 0:
            print(f"  {it + 1} iterations, {bad} mismatching batches so far, {time.time() - t0:.0f} s", flush=True)
    print(f"{iters} iterations, {bad} mismatching batches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
