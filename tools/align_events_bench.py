#!/usr/bin/env python3
"""The drop-in boundary with HOST buffers (what the reference's align_db hook calls): sfa_align_events on event tables (AoS,
24 bytes per event) and sfa_align_batch on packed means, per batch size, next to the device-resident stage (run on the GPU box).
The marshalling of the Python list of tables into the pointer array is done ONCE, outside the timed calls."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sigfish_amd as S  # noqa: E402
from sigfish_amd import _lib, synth  # noqa: E402


def main():
    torch.zeros(1, device="cuda")
    L = _lib.load()
    ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=0)
    for n in (int(a) for a in (sys.argv[1:] or ["512", "4096", "100000"])):
        q, q_off, _ = synth.make_reads(ref, n, qlen=250, seed=5)
        lens = (q_off[1:] - q_off[:-1]).astype(np.int64)
        # one big AoS block: read i's table starts at ev_off[i], its window 50 events in
        ev_off = np.concatenate([[0], np.cumsum(lens + 60)]).astype(np.int64)
        ev = np.zeros(int(ev_off[-1]), S.EVENT_DTYPE)
        for i in range(n):
            ev["mean"][ev_off[i] + 50:ev_off[i] + 50 + lens[i]] = q[q_off[i]:q_off[i + 1]]
        base = ev.ctypes.data
        ptrs = (C.c_void_p * n)(*[base + int(o) * S.EVENT_DTYPE.itemsize for o in ev_off[:-1]])
        nev = (lens + 60).astype(np.int64)
        qs = np.full(n, 50, np.int64)
        qe = (50 + lens).astype(np.int64)
        rows = np.zeros(n, S.RESULT_DTYPE)
        with S.Aligner(ref, flag) as al:
            want = al.align_db(q, q_off)

            def call():
                rc = L.sfa_align_events(al._h, C.cast(ptrs, C.POINTER(C.POINTER(_lib.SfaEvent))), nev.ctypes.data_as(_lib.i64p), qs.ctypes.data_as(_lib.i64p), qe.ctypes.data_as(_lib.i64p), n,
                                        rows.ctypes.data_as(C.c_void_p))
                assert rc == 0, rc
            call()
            reps = 5 if n < 50000 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            t_ev = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                al.align_db(q, q_off)
            t_db = (time.perf_counter() - t0) / reps
            d_q = torch.from_numpy(q).cuda()
            out = torch.zeros(n * S.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
            al.align_db_device(d_q.data_ptr(), q_off, n, out.data_ptr(), sync=True)
            t0 = time.perf_counter()
            for _ in range(reps):
                al.align_db_device(d_q.data_ptr(), q_off, n, out.data_ptr(), sync=True)
            t_dev = (time.perf_counter() - t0) / reps
        print(f"reads {n}: sfa_align_events (AoS event tables) {t_ev * 1e3:.2f} ms = {n / t_ev:.0f} reads/s; "
              f"sfa_align_batch (packed host means) {t_db * 1e3:.2f} ms = {n / t_db:.0f} reads/s; device-resident {t_dev * 1e3:.2f} ms = {n / t_dev:.0f} reads/s; "
              f"rows equal: {rows.tobytes() == want.tobytes()}", flush=True)


if __name__ == "__main__":
    main()
