import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import sigfish_amd as S
from sigfish_amd import synth
from oracle import oracle as O
ref, flag, _, _, _ = synth.workload("ncov_r9_dna_q250", n_reads=8, seed=0)
oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
for n in (64, 256, 512, 1024, 2048, 4096):
    q, q_off, _ = synth.make_reads(ref, n, qlen=250, seed=n)
    res = {}
    for seg in (1, 0):
        with S.Aligner(ref, flag) as al:
            al.set_option("column_segments", seg)
            al.align_db(q, q_off)
            t = []
            for _ in range(5):
                t0 = time.perf_counter(); rows = al.align_db(q, q_off); t.append(time.perf_counter() - t0)
            p = al.profile()
            res[seg] = rows
            print(f"reads {n:5d} column_segments={seg}: {min(t)*1e3:6.2f} ms per batch, fill {p['fill_ms']:.2f} trace {p['trace_ms']:.2f}  segments {p['n_segments']} chunks {p['n_chunks']} reruns {p['segment_reruns']}", flush=True)
    assert res[0].tobytes() == res[1].tobytes()
    if n <= 256:
        want = O.align_batch(q, q_off, oref, flag, threads=16)
        assert res[0].tobytes() == want.tobytes()
print("rows identical with and without segments")
