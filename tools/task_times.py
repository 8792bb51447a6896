#!/usr/bin/env python3
"""When does every wave-task of the fill start and end, and on which SIMD?  Needs a -DSFA_TASK_TIMES build of the library:
    make -C sigfish_amd/csrc NAME=libsfa_times.so EXTRA=-DSFA_TASK_TIMES
    SFA_LIB=sigfish_amd/lib/libsfa_times.so python tools/task_times.py [reads ...]
Prints, per batch size: the fill time, the spread of task durations, and per SIMD the times at which its k-th task ended
(mean over SIMDs) -- the picture behind "a batch takes longer than its share of a big one" (DESIGN.md section 7)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import sigfish_amd as S
from sigfish_amd import _lib, synth


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 12288, 16384, 100000]
    torch.cuda.set_device(0)
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=0)
    al = S.Aligner(ref, flag, device=0)
    L = _lib.load()
    L.sfa_debug_task_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.sfa_debug_task_times.restype = C.c_int64
    for n in sizes:
        q, q_off, _ = synth.make_reads(ref, n, qlen=250, seed=5, short_frac=0.0)
        dq = torch.from_numpy(q).cuda()
        dout = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            al.align_db_device(dq.data_ptr(), q_off, n, dout.data_ptr(), sync=True)
        p = al.profile()
        buf = np.zeros((p["n_tasks"], 3), np.uint64)
        got = L.sfa_debug_task_times(al._h, buf.ctypes.data, p["n_tasks"])
        assert got == p["n_tasks"], got
        t0 = buf[:, 0].min()
        st = (buf[:, 0] - t0).astype(np.float64) / 1e5  # ms (100 MHz ticks)
        en = (buf[:, 1] - t0).astype(np.float64) / 1e5
        simd = buf[:, 2].astype(np.int64)
        print(f"reads {n}: tasks {len(buf)}, fill {p['fill_ms']:.2f} ms; task duration ms min/mean/max "
              f"{(en - st).min():.2f}/{(en - st).mean():.2f}/{(en - st).max():.2f}; last end {en.max():.2f}")
        per = {}
        for s, a, b in zip(simd, st, en):
            per.setdefault(s, []).append((a, b))
        ks = max(len(v) for v in per.values())
        counts = np.bincount([len(v) for v in per.values()])
        print("  SIMDs by number of tasks:", {k: int(c) for k, c in enumerate(counts) if c})
        ends = np.full((len(per), ks), np.nan)
        for i, v in enumerate(per.values()):
            e = sorted(b for _, b in v)
            ends[i, :len(e)] = e
        if ks <= 12:
            print("  k-th task end on a SIMD, mean over SIMDs (ms):", " ".join(f"{x:.2f}" for x in np.nanmean(ends, axis=0)))
        last = np.nanmax(ends, axis=1)
        print(f"  last end per SIMD: min {last.min():.2f} p10 {np.percentile(last, 10):.2f} median {np.median(last):.2f} "
              f"p90 {np.percentile(last, 90):.2f} max {last.max():.2f}")
        # how busy is the chip over time: tasks in flight per 0.5 ms
        edges = np.arange(0, en.max() + 0.5, 0.5)
        infl = [(np.minimum(en, b) - np.maximum(st, a)).clip(0).sum() / 0.5 for a, b in zip(edges[:-1], edges[1:])]
        print("  tasks in flight per 0.5 ms bin (last 16):", " ".join(f"{x:.0f}" for x in infl[-16:]))
    al.close()


if __name__ == "__main__":
    main()
