#!/usr/bin/env python3
"""Interleaved A/B of two option sets of the alignment stage in ONE process (run on an MI355X box): per batch size, `--rounds`
rounds of (A: `--steps` steps, B: `--steps` steps), device-resident queries, median step time per side; rows of A and B
compared byte for byte.

    python tools/ab_opts.py --workload ncov_r9_dna_q250 --reads 8192,16384,100000 --a fused_trace=0 --b fused_trace=2"""
import argparse
import os
import re
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ncov_r9_dna_q250")
    ap.add_argument("--reads", default="8192,16384,100000")
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    import torch

    import sigfish_amd as S
    from sigfish_amd import synth
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    ref, flag, _, _, _ = synth.workload(args.workload, n_reads=8, seed=0)
    qlen = int(re.search(r"_q(\d+)$", args.workload).group(1))
    sides = {}
    for name, spec in (("A", args.a), ("B", args.b)):
        al = S.Aligner(ref, flag, device=0)
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            al.set_option(k, int(v))
        sides[name] = al
    print(f"workload {args.workload}; A: {args.a or 'defaults'}; B: {args.b or 'defaults'}; median of {args.rounds} x {args.steps} steps per side, interleaved")
    for n in (int(x) for x in args.reads.split(",")):
        q, q_off, _ = synth.make_reads(ref, n, qlen=qlen, seed=1000)
        d_q = torch.from_numpy(q).to(dev)
        outs = {k: torch.zeros(n * S.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for k in sides}
        times = {k: [] for k in sides}
        fills = {k: [] for k in sides}
        tasks = {}
        for k, al in sides.items():  # warm-up
            al.align_db_device(d_q.data_ptr(), q_off, n, outs[k].data_ptr(), sync=True)
        for _ in range(args.rounds):
            for k, al in sides.items():
                for _ in range(args.steps):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    al.align_db_device(d_q.data_ptr(), q_off, n, outs[k].data_ptr(), sync=True)
                    times[k].append((time.perf_counter() - t0) * 1e3)
                    p = al.profile()
                    fills[k].append(p["fill_ms"] + p["trace_ms"])
                    tasks[k] = p["n_tasks"]
        same = bool(torch.equal(outs["A"], outs["B"]))
        a, b = statistics.median(times["A"]), statistics.median(times["B"])
        print(f"reads {n}: A {a:.3f} ms (device {statistics.median(fills['A']):.3f}, {tasks['A']} tasks)  "
              f"B {b:.3f} ms (device {statistics.median(fills['B']):.3f}, {tasks['B']} tasks)  B/A {b / a:.4f}  rows equal: {same}", flush=True)
    for al in sides.values():
        al.close()


if __name__ == "__main__":
    main()
