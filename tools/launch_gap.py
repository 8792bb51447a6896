#!/usr/bin/env python3
"""launch_gap.py -- how the time of one fill launch depends on what the GPU did just before (run on an MI355X box):
back-to-back batches, batches behind a concurrently spinning kernel, batches after 50 ms of idle.  Device-resident
queries, 16-lane shapes, fill time from the library's own HIP events."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.set_device(0)
import sigfish_amd as S
from sigfish_amd import synth
ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=0)
for n in (8192, 16384, 32768, 100000):
    q, q_off, _ = synth.make_reads(ref, n, qlen=250, seed=n)
    d_q = torch.from_numpy(q).cuda()
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    with S.Aligner(ref, flag) as al:
        al.set_option("lane_widening", 1)
        for mode in ("plain", "spin", "plain", "spin", "idle50ms"):
            res = []
            for it in range(6):
                if mode == "spin":
                    torch.cuda._sleep(int(6e7))  # ~25-30 ms of spinning on torch's stream: the GPU never idles around the fill
                if mode == "idle50ms":
                    time.sleep(0.05)
                al.align_db_device(d_q.data_ptr(), q_off, n, d_out.data_ptr(), sync=True)
                p = al.profile()
                res.append(p["fill_ms"])
                if mode == "spin":
                    torch.cuda.synchronize()
            print(f"reads {n:6d} {mode:9s} fill ms: " + " ".join(f"{x:7.3f}" for x in res), flush=True)
