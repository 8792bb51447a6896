#!/usr/bin/env python3
"""Timeline of the pipelined pass 1 over row strips (queries beyond 2048 events): when does every (job, read, strip) wave start
and end, on which SIMD, how many are resident over time.  Needs a -DSFA_TASK_TIMES build of the library:
    make -C sigfish_amd/csrc NAME=libsfa_times.so EXTRA=-DSFA_TASK_TIMES
    SFA_LIB=sigfish_amd/lib/libsfa_times.so python tools/strip_task_times.py [qlen [reads]]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import sigfish_amd as S
from sigfish_amd import _lib, synth


def main():
    qlen = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000 * 250 // qlen
    torch.cuda.set_device(0)
    ref, flag, _, _, _ = synth.workload(f"ncov_r9_dna_q{qlen}", n_reads=8, seed=0)
    al = S.Aligner(ref, flag, device=0)
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        al.set_option(k, int(v))
    L = _lib.load()
    L.sfa_debug_task_times_long.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.sfa_debug_task_times_long.restype = C.c_int64
    q, q_off, _ = synth.make_reads(ref, n, qlen=qlen, seed=5, short_frac=0.0)
    dq = torch.from_numpy(q).cuda()
    dout = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    import time
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.time()
        try:
            al.align_db_device(dq.data_ptr(), q_off, n, dout.data_ptr(), sync=True)
        except Exception as e:  # experiment builds with deliberately wrong rows
            print("  (align failed:", str(e)[:80], ")")
        torch.cuda.synchronize()
        step = (time.time() - t) * 1e3
    cap = n * 2 * 8
    buf = np.zeros((cap, 3), np.uint64)
    got = L.sfa_debug_task_times_long(al._h, buf.ctypes.data, cap)
    buf = buf[:got]
    t0 = buf[:, 0].min()
    st = (buf[:, 0] - t0).astype(np.float64) / 1e5
    en = (buf[:, 1] - t0).astype(np.float64) / 1e5
    simd = (buf[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    sidx = (buf[:, 2] >> np.uint64(32)).astype(np.int64)
    d = en - st
    print(f"q {qlen}, reads {n}: step {step:.1f} ms, strip tasks {got}, pass 1 spans {en.max():.2f} ms")
    print(f"  task duration ms: min {d.min():.2f} p10 {np.percentile(d, 10):.2f} median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f}")
    for s in range(sidx.max() + 1):
        m = sidx == s
        print(f"  strip {s}: {m.sum()} tasks, duration median {np.median(d[m]):.2f} ms, start median {np.median(st[m]):.2f}")
    xcd = simd // (8 * 2 * 16 * 4)
    for x in range(8):
        m = xcd == x
        if m.any():
            print(f"  XCD {x}: {m.sum()} tasks, last end {en[m].max():.2f} ms, mean duration {d[m].mean():.2f}")
    edges = np.arange(0, en.max() + 2.0, 2.0)
    infl = [(np.minimum(en, b) - np.maximum(st, a)).clip(0).sum() / 2.0 for a, b in zip(edges[:-1], edges[1:])]
    print("  waves resident per 2 ms bin:", " ".join(f"{x:.0f}" for x in infl))
    per_simd = np.bincount(simd)
    per_simd = per_simd[per_simd > 0]
    print(f"  tasks per SIMD: min {per_simd.min()} median {int(np.median(per_simd))} max {per_simd.max()} over {len(per_simd)} SIMDs")
    al.close()


if __name__ == "__main__":
    main()
