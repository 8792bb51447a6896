"""The host pre-DP stages (sigfish_amd/csrc/host/events.cpp: event detection, adaptor / poly-A query start, query window,
z-normalisation) against the COMPILED REFERENCE on hundreds of synthetic step signals -- the fixtures alone are 13 reads.
Runs where oracle/_ref exists (the build container); the device-side stages are in turn checked against these host stages
(tests/test_raw_gpu.py, tools/fuzz_raw_gpu.py)."""
import os
import subprocess
import zlib

import numpy as np
import pytest

import sigfish_amd as S
from oracle import oracle as O
from tests.util import GOLD, write_blow5

pytestmark = pytest.mark.skipif(not os.path.exists(O.REF_DRIVER), reason="oracle/_ref not built (no /root/reference)")


def _signal(rng, n):
    dwell = rng.integers(2, int(rng.integers(4, 40)), size=n // 2 + 2)
    levels = rng.normal(rng.uniform(300, 700), rng.uniform(20, 120), size=len(dwell))
    x = np.repeat(levels, dwell)[:n]
    if len(x) < n:
        x = np.concatenate([x, np.full(n - len(x), 500.0)])
    x = x + rng.normal(0, rng.uniform(0.5, 15), size=n)
    if rng.integers(0, 6) == 0:
        x = np.round(x / 8) * 8
    return np.clip(np.round(x), -2000, 4000).astype(np.int16)


@pytest.mark.parametrize("case", [("dna", []), ("dna_q100_p10", ["-q", "100", "-p", "10"]), ("dna_end", ["--from-end"]),
                                  ("rna", ["--rna"]), ("rna_pauto", ["--rna", "-p", "-1"]), ("rna_q500", ["--rna", "-q", "500"])])
def test_random_signals(case, tmp_path):
    name, args = case
    rna = "--rna" in args
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    reads = []
    for k in range(150):
        # (very short reads are outside the reference's domain: it asserts in trim_raw_by_mad, src/events.c:246, below ~200
        # samples, and a 358-sample read made it abort with heap corruption; a real read has thousands of samples)
        n = int(rng.choice([1000, 1500, 2500, 4000, 9000, 20000], p=[.05, .1, .2, .3, .25, .1]) * rng.uniform(0.6, 1.4))
        reads.append((f"read{k}", float(rng.choice([2048.0, 8192.0])), float(rng.integers(-20, 40)), float(rng.uniform(700, 1500)), 4000.0,
                      _signal(rng, n)))
    blow5 = str(tmp_path / "rnd.blow5")
    # half of the cases as real files are written (zlib records, svb-zd signals): the reference reads them through slow5lib,
    # our stages through the library's own reader -- same events means same decoded samples
    compress = name in ("dna_q100_p10", "rna", "rna_q500")
    write_blow5(blow5, reads, attrs=(("experiment_type", "rna" if rna else "genomic_dna"), ("sequencing_kit", "unknown")), compress=compress)
    if compress:
        back = list(S.Blow5File(blow5))
        assert len(back) == len(reads) and all(np.array_equal(b[2], r[5]) and b[0] == r[0] for b, r in zip(back, reads))
    k = 5 if rna else 6
    fasta = os.path.join(GOLD, "data", "rnasequin_sequences_2.4.fa" if rna else "nCoV-2019.reference.fasta")
    dump = str(tmp_path / "dump.bin")
    paf = subprocess.run([O.REF_DRIVER, "--model", os.path.join(GOLD, "models", f"syn{k}.f32"), "--kmer", str(k), "--dump", dump, *args, fasta, blow5],
                         check=True, capture_output=True).stdout.decode()
    d = O.parse_dump(dump)
    prefix = int(args[args.index("-p") + 1]) if "-p" in args else 50
    query = int(args[args.index("-q") + 1]) if "-q" in args else 250
    flag = (S.RNA if rna else 0) | (S.END if "--from-end" in args else 0)
    assert len(d["reads"]) == len(reads)
    checked = 0
    for want, (rid, dig, off, rng_, rate, raw) in zip(d["reads"], reads):
        meta = dict(digitisation=dig, offset=off, range=rng_)
        ev = S.detect_events(raw, meta, rna)
        if want["valid"]:  # a read the reference drops has its event count zeroed (et.n = 0, src/sigfish.c:470-476)
            assert want["n_events"] == len(ev), rid
        keep, a, b = (False, 0, 0)
        if len(ev):
            keep, a, b = S.select_query(ev, raw, meta, prefix, query, flag, 0)
        assert keep == want["valid"], rid
        if keep:
            assert (a, b) == (want["qstart"], want["qend"]), rid
            assert np.array_equal(ev["mean"][a:b].view(np.uint32), want["query"].view(np.uint32)), rid
            assert int(ev["start"][a]) == want["ev_start_first"] and int(ev["start"][b - 1]) == want["ev_start_last"]
            assert np.float32(ev["length"][b - 1]) == want["ev_len_last"]
            checked += 1
    assert checked > 100
    # the alignment and the PAF text for the same reads: our CPU restatement of the alignment stage (the oracle the GPU
    # tests compare against) on the host stages' queries, printed by sfa_paf_row, against what the reference printed
    valid = [(w, r) for w, r in zip(d["reads"], reads) if w["valid"]]
    q = np.concatenate([w["query"] for w, _ in valid])
    q_off = np.concatenate([[0], np.cumsum([len(w["query"]) for w, _ in valid])]).astype(np.int64)
    rows = O.align_batch(q, q_off, d["ref"], d["flag"], threads=8)
    lines = []
    for (w, (rid, dig, off, rng_, rate, raw)), r in zip(valid, rows):
        for f in ("rid", "pos_st", "pos_end", "strand", "mapq"):
            assert r[f] == w[f], (rid, f)
        assert np.float32(r["score"]) == w["score"] and (np.float32(r["score2"]) == w["score2"] or (np.isinf(r["score2"]) and np.isinf(w["score2"])))
        end_raw = int(np.float32(np.float32(w["ev_start_last"]) + w["ev_len_last"]))  # u64 + float in C: fp32 arithmetic
        lines.append(S.paf_row(r, rid, d["ref"].names[int(r["rid"])], w["ev_start_first"], end_raw, (w["qend"] - 1) - w["qstart"], len(raw),
                               int(d["ref"].seq_lengths[int(r["rid"])])))
    assert "".join(lines) == paf


@pytest.mark.parametrize("case", [("dna_sam", ["--sam"]), ("rna_sam", ["--rna", "--sam"]), ("rna_fullref_sam", ["--rna", "--full-ref", "--sam", "-q", "400"])])
def test_random_signals_sam(case, tmp_path):
    """The SAM writer's warp-path recovery (band between the winner's start and end columns instead of the reference's
    full-matrix traceback) on random alignments: the reference's own rows in, its SAM text out, byte for byte."""
    name, args = case
    rna = "--rna" in args
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    reads = []
    for k in range(80):
        n = int(rng.choice([2500, 4000, 9000], p=[.3, .4, .3]) * rng.uniform(0.7, 1.4))
        reads.append((f"read{k}", 8192.0, float(rng.integers(-20, 40)), float(rng.uniform(700, 1500)), 4000.0, _signal(rng, n)))
    blow5 = str(tmp_path / "rnd.blow5")
    write_blow5(blow5, reads, attrs=(("experiment_type", "rna" if rna else "genomic_dna"), ("sequencing_kit", "unknown")))
    k = 5 if rna else 6
    fasta = os.path.join(GOLD, "data", "rnasequin_sequences_2.4.fa" if rna else "nCoV-2019.reference.fasta")
    dump = str(tmp_path / "dump.bin")
    out = subprocess.run([O.REF_DRIVER, "--model", os.path.join(GOLD, "models", f"syn{k}.f32"), "--kmer", str(k), "--dump", dump, *args, fasta, blow5],
                         check=True, capture_output=True).stdout.decode()
    want = [l + "\n" for l in out.split("\n") if l and not l.startswith("@")]
    d = O.parse_dump(dump)
    query = int(args[args.index("-q") + 1]) if "-q" in args else 250
    flag = (S.RNA if rna else 0) | (S.REF if "--full-ref" in args else 0)
    levels = np.fromfile(os.path.join(GOLD, "models", f"syn{k}.f32"), np.float32)
    ref = S.RefModel.from_fasta(fasta, levels, k, flag, query)
    got = []
    for w, (rid, dig, off, rng_, rate, raw) in zip(d["reads"], reads):
        if not w["valid"]:
            continue
        meta = dict(digitisation=dig, offset=off, range=rng_)
        ev = S.detect_events(raw, meta, rna)
        keep, a, b = S.select_query(ev, raw, meta, 50, query, flag, 0)
        assert keep and (a, b) == (w["qstart"], w["qend"])
        row = np.zeros(1, S.RESULT_DTYPE)[0]
        for f in ("rid", "pos_st", "pos_end", "score", "score2", "strand", "mapq"):
            row[f] = w[f]
        row["valid"] = 1
        arr = ref.forward[int(row["rid"])] if row["strand"] == ord("+") else ref.reverse[int(row["rid"])]
        got.append(S.sam_row(row, rid, ref.names[int(row["rid"])], ev, a, b, arr, int(ref.st_offset[int(row["rid"])]), flag))
    assert len(got) > 60 and got == want


@pytest.mark.parametrize("case", [("dna", [], 6), ("dna_q100", ["-q", "100"], 6), ("rna", ["--rna"], 5), ("rna_q500", ["--rna", "-q", "500"], 5),
                                  ("rna_full", ["--rna", "--full-ref"], 5)])
def test_random_fasta_gen_ref(case, tmp_path):
    """gen_ref (src/genref.c:86-241) on random contigs -- lengths around the RNA slice limits, very short contigs --
    against sfa_gen_ref_record: lengths, offsets and every float of the forward / reverse arrays."""
    name, args, k = case
    rna = "--rna" in args
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    query = int(args[args.index("-q") + 1]) if "-q" in args else 250
    fasta = str(tmp_path / "rnd.fa")
    with open(fasta, "w") as f:
        for c in range(12):
            # (with --full-ref a one-k-mer contig makes the reference's aligner assert, src/sigfish.c:611: start at 40 there)
            n = int(rng.choice([40 if "--full-ref" in args else k, k + 36, 40, query, query + k, int(query * 1.5) + k - 1, int(query * 1.5) + k, 2000, 7000]))
            seq = "".join(rng.choice(list("ACGT"), size=n))
            f.write(f">contig{c} some description\n")
            for i in range(0, n, 61):
                f.write(seq[i:i + 61] + "\n")
    dump = str(tmp_path / "dump.bin")
    blow5 = os.path.join(GOLD, "data", "sequin_rna.blow5" if rna else "sp1_dna.blow5")
    subprocess.run([O.REF_DRIVER, "--model", os.path.join(GOLD, "models", f"syn{k}.f32"), "--kmer", str(k), "--dump", dump, *args, fasta, blow5],
                   check=True, capture_output=True)
    want = O.parse_dump(dump)["ref"]
    flag = (S.RNA if rna else 0) | (S.REF if "--full-ref" in args else 0)
    levels = np.fromfile(os.path.join(GOLD, "models", f"syn{k}.f32"), np.float32)
    got = S.RefModel.from_fasta(fasta, levels, k, flag, query)
    assert list(got.names) == list(want.names)
    assert np.array_equal(got.ref_lengths, want.ref_lengths) and np.array_equal(got.seq_lengths, want.seq_lengths)
    assert np.array_equal(got.st_offset, want.st_offset)
    for i in range(got.num_ref):
        assert np.array_equal(got.forward[i].view(np.uint32), want.forward[i].view(np.uint32)), i
        if not rna:
            assert np.array_equal(got.reverse[i].view(np.uint32), want.reverse[i].view(np.uint32)), i
