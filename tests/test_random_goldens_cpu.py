"""tests/golden/random (synthetic step signals in compressed BLOW5 files + what the compiled reference printed for them)
through the CPU side of the repository: own BLOW5 reader -> host event detection / query window -> the oracle's alignment
-> PAF / SAM writers.  Runs anywhere; the same fixtures go through the GPU command line in tests/test_cli_gpu.py."""
import glob
import os

import numpy as np
import pytest

import sigfish_amd as S
from oracle import oracle as O
from tests.util import GOLD

CASES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLD, "random", "*.args")))


@pytest.mark.parametrize("name", CASES)
def test_host_pipeline_reproduces_reference_output(name):
    k, fasta, blow5, *args = open(os.path.join(GOLD, "random", name + ".args")).read().split("\n")
    k = int(k)
    rna, sam = "--rna" in args, "--sam" in args
    flag = (S.RNA if rna else 0) | (S.REF if "--full-ref" in args else 0) | (S.DTW if "--dtw-std" in args else 0) | \
        (S.END if "--from-end" in args else 0)
    prefix = int(args[args.index("-p") + 1]) if "-p" in args else 50
    query = int(args[args.index("-q") + 1]) if "-q" in args else 250
    levels = np.fromfile(os.path.join(GOLD, "models", f"syn{k}.f32"), np.float32)
    ref = S.RefModel.from_fasta(os.path.join(GOLD, "data", fasta), levels, k, flag, query)
    kept = []
    for rid, meta, raw in S.Blow5File(os.path.join(GOLD, "random", blow5)):
        ev = S.detect_events(raw, meta, rna)
        keep, a, b = S.select_query(ev, raw, meta, prefix, query, flag, 0) if len(ev) else (False, 0, 0)
        if keep:
            kept.append((rid, len(raw), ev, a, b))
    q = np.concatenate([ev["mean"][a:b] for _, _, ev, a, b in kept])
    q_off = np.concatenate([[0], np.cumsum([b - a for _, _, _, a, b in kept])]).astype(np.int64)
    oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    rows = O.align_batch(q, q_off, oref, flag, threads=8)
    lines = []
    for (rid, nraw, ev, a, b), r in zip(kept, rows):
        ci = int(r["rid"])
        if sam:
            arr = ref.forward[ci] if r["strand"] == ord("+") else ref.reverse[ci]
            lines.append(S.sam_row(r, rid, ref.names[ci], ev, a, b, arr, int(ref.st_offset[ci]), flag))
        else:
            end_raw = int(np.float32(np.float32(ev["start"][b - 1]) + ev["length"][b - 1]))
            lines.append(S.paf_row(r, rid, ref.names[ci], int(ev["start"][a]), end_raw, (b - 1) - a, nraw, int(ref.seq_lengths[ci])))
    want = [l + "\n" for l in open(os.path.join(GOLD, "random", name + ".out")).read().split("\n") if l and not l.startswith("@")]
    assert lines == want
