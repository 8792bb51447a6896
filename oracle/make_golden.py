#!/usr/bin/env python3
"""make_golden.py -- TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/ from the COMPILED REFERENCE.

Run in the build container (needs /root/reference and `make -C oracle ref`):
    python oracle/make_golden.py

What it writes (all small, all data -- no reference source text):
  tests/golden/data/      the reference's own test inputs (2 FASTA, 2 BLOW5 fixtures), byte-for-byte
  tests/golden/models/    seeded synthetic k-mer level tables (float32), k=5 and k=6
  tests/golden/cases/     for every flag combination: <case>.out (PAF/SAM text printed by the reference's
                          output_db) and <case>.npz (normalised query events, per-read metadata, aln_t rows,
                          sha256 + shape of the reference event arrays produced by the reference's gen_ref)
  tests/golden/kernel_vectors.npz   cdtw.c-level vectors (subsequence / std_dtw last rows, window minima,
                          traceback start columns) computed by the reference's cdtw.c through ctypes
"""
import hashlib
import itertools
import os
import random
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REFTEST = "/root/reference/test"
GOLD = os.path.join(ROOT, "tests", "golden")


def synth_levels(k, seed, mu, sd):
    """SURVEY.md Appendix A recipe: random.seed(seed); gauss(mu, sd) printed with 4 decimals, lexicographic ACGT."""
    random.seed(seed)
    vals = [float("%.4f" % random.gauss(mu, sd)) for _ in itertools.product("ACGT", repeat=k)]
    return np.array(vals, dtype=np.float32)


CASES = [
    # name, fasta, blow5, model k, args
    ("dna_default", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, []),
    ("dna_t1", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, ["-t", "1"]),
    ("dna_from_end", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, ["--from-end"]),
    ("dna_q100", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, ["-q", "100"]),
    ("dna_q300_p10", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, ["-q", "300", "-p", "10"]),
    ("dna_sam", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, ["--sam"]),
    ("dna_q700", "nCoV-2019.reference.fasta", "sp1_dna.blow5", 6, ["-q", "700"]),
    ("rna_default", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna"]),
    ("rna_q500_pauto", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "-q", "500", "-p", "-1"]),
    ("rna_full_ref", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "--full-ref"]),
    ("rna_full_ref_dtw_std", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5,
     ["--rna", "--full-ref", "--dtw-std"]),
    ("rna_dtw_std", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "--dtw-std"]),
    ("rna_from_end", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "--from-end"]),
    ("rna_invert", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "--invert"]),
    ("rna_sam", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "--sam"]),
    ("rna_q1000", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "-q", "1000", "--full-ref"]),
    ("rna_q2000_sam", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "-q", "2000", "--full-ref", "--sam"]),
    # queries beyond 2048 events (row strips on the GPU side): 7 of the 8 reads have more than 2048 + 50 events
    ("rna_q2500", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "-q", "2500"]),
    ("rna_q4200_full_sam", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "-q", "4200", "--full-ref", "--sam"]),
    ("rna_q3000_full_dtw_std", "rnasequin_sequences_2.4.fa", "sequin_rna.blow5", 5, ["--rna", "-q", "3000", "--full-ref", "--dtw-std"]),
]


def sha(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, "<f4").tobytes())
    return h.hexdigest()


def run_case(name, fasta, blow5, k, args):
    dump = f"/tmp/_golden_{name}.dump"
    cmd = [O.REF_DRIVER, "--model", os.path.join(GOLD, "models", f"syn{k}.f32"), "--kmer", str(k), "--dump", dump,
           *args, os.path.join(GOLD, "data", fasta), os.path.join(GOLD, "data", blow5)]
    out = subprocess.run(cmd, check=True, capture_output=True).stdout
    with open(os.path.join(GOLD, "cases", name + ".out"), "wb") as f:
        f.write(out)
    d = O.parse_dump(dump)
    os.remove(dump)
    reads = d["reads"]
    valid = [r for r in reads if r["valid"]]
    q = np.concatenate([r["query"] for r in valid]) if valid else np.zeros(0, np.float32)
    q_off = np.cumsum([0] + [len(r["query"]) for r in valid]).astype(np.int64)
    ref = d["ref"]
    np.savez_compressed(
        os.path.join(GOLD, "cases", name + ".npz"),
        flag=np.int32(d["flag"]), k=np.int32(k), args=np.array(args, dtype="U32"), fasta=fasta, blow5=blow5,
        read_ids=np.array([r["read_id"] for r in reads], dtype="U64"),
        read_valid=np.array([r["valid"] for r in reads], bool),
        len_raw=np.array([r["len_raw"] for r in reads], np.int64),
        n_events=np.array([r["n_events"] for r in reads], np.int64),
        qstart=np.array([r["qstart"] for r in reads], np.int64),
        qend=np.array([r["qend"] for r in reads], np.int64),
        ev_start_first=np.array([r["ev_start_first"] for r in valid], np.uint64),
        ev_start_last=np.array([r["ev_start_last"] for r in valid], np.uint64),
        ev_len_last=np.array([r["ev_len_last"] for r in valid], np.float32),
        queries=q, q_off=q_off,
        rid=np.array([r["rid"] for r in valid], np.int32),
        pos_st=np.array([r["pos_st"] for r in valid], np.int32),
        pos_end=np.array([r["pos_end"] for r in valid], np.int32),
        score=np.array([r["score"] for r in valid], np.float32),
        score2=np.array([r["score2"] for r in valid], np.float32),
        strand=np.array([r["strand"] for r in valid], np.int8),
        mapq=np.array([r["mapq"] for r in valid], np.uint8),
        ref_names=np.array(ref.names, dtype="U64"), ref_lengths=ref.ref_lengths, ref_seq_lengths=ref.seq_lengths,
        ref_st_offset=ref.st_offset, fwd_sha256=sha(ref.forward),
        rev_sha256=sha(ref.reverse) if ref.reverse is not None else "",
    )
    print(f"{name}: {len(valid)}/{len(reads)} reads, {ref.num_ref} contigs")


def kernel_vectors():
    """cdtw.c-level known answers from the reference build: random + tie-heavy + edge shapes."""
    rng = np.random.default_rng(20240611)
    shapes = [(1, 1), (1, 7), (7, 1), (2, 2), (5, 40), (25, 60), (33, 33), (16, 200), (40, 17), (64, 129), (100, 260)]
    out = {}
    idx = 0
    for (n, m) in shapes:
        for mode in ("gauss", "quant", "const"):
            if mode == "gauss":
                x = rng.normal(size=n).astype(np.float32)
                y = rng.normal(size=m).astype(np.float32)
            elif mode == "quant":  # quarter-integers: exact sums -> many exact ties
                x = (rng.integers(-4, 5, size=n) / 4).astype(np.float32)
                y = (rng.integers(-4, 5, size=m) / 4).astype(np.float32)
            else:
                x = np.full(n, 0.5, np.float32)
                y = np.full(m, 0.5, np.float32)
            cs = O.ref_subsequence(x, y)
            cd = O.ref_std_dtw(x, y)
            starts_s = np.array([O.ref_subsequence_path(cs, j)[1][0] for j in range(m)], np.int32)
            starts_d = np.array([O.ref_subsequence_path(cd, j)[1][0] for j in range(m)], np.int32)
            out[f"x{idx}"], out[f"y{idx}"] = x, y
            out[f"sub_last{idx}"], out[f"std_last{idx}"] = cs[-1].copy(), cd[-1].copy()
            out[f"sub_start{idx}"], out[f"std_start{idx}"] = starts_s, starts_d
            if n * m <= 1200:
                out[f"sub_full{idx}"], out[f"std_full{idx}"] = cs, cd
                px, py = O.ref_subsequence_path(cs, m - 1)
                out[f"sub_px{idx}"], out[f"sub_py{idx}"] = px.astype(np.int32), py.astype(np.int32)
            idx += 1
    out["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(GOLD, "kernel_vectors.npz"), **out)
    print(f"kernel vectors: {idx}")


def eval_goldens():
    """`sigfish eval truth.paf test.paf` (src/eval.c) printed by the reference build for the fixture PAFs."""
    os.makedirs(os.path.join(GOLD, "eval"), exist_ok=True)
    pairs = [("dna", "sp1_dna.minimap2.paf", "dna_default.out", []), ("rna", "sequin_rna.minimap2.paf", "rna_default.out", []),
             ("rna_full_ref", "sequin_rna.minimap2.paf", "rna_full_ref.out", []),
             ("rna_tid_only", "sequin_rna.minimap2.paf", "rna_default.out", ["--tid-only"]),
             ("rna_nosec", "sequin_rna.minimap2.paf", "rna_q500_pauto.out", ["--secondary", "no"])]
    for name, truth, test, extra in pairs:
        cmd = [O.REF_DRIVER, "eval", *extra, os.path.join(GOLD, "data", truth), os.path.join(GOLD, "cases", test)]
        r = subprocess.run(cmd, check=True, capture_output=True)
        with open(os.path.join(GOLD, "eval", name + ".txt"), "wb") as f:
            f.write(r.stdout)
        with open(os.path.join(GOLD, "eval", name + ".args"), "w") as f:
            f.write("\n".join([truth, test] + extra))
    print("eval goldens:", len(pairs))


def random_goldens():
    """Synthetic step signals (compressed BLOW5, 40 reads each) through the reference's batch loop: more end-to-end
    known answers for the command line than the 13 reads of the reference's own fixtures."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import write_blow5  # noqa: E402
    rd = os.path.join(GOLD, "random")
    os.makedirs(rd, exist_ok=True)

    def signal(rng, n):
        dwell = rng.integers(2, int(rng.integers(4, 40)), size=n // 2 + 2)
        levels = rng.normal(rng.uniform(300, 700), rng.uniform(20, 120), size=len(dwell))
        x = np.repeat(levels, dwell)[:n]
        if len(x) < n:
            x = np.concatenate([x, np.full(n - len(x), 500.0)])
        x = x + rng.normal(0, rng.uniform(0.5, 15), size=n)
        return np.clip(np.round(x), -2000, 4000).astype(np.int16)

    for kind, k, fasta, cases in (("dna", 6, "nCoV-2019.reference.fasta", [("rnd_dna", []), ("rnd_dna_sam", ["--sam"]), ("rnd_dna_end_q100", ["--from-end", "-q", "100"])]),
                                  ("rna", 5, "rnasequin_sequences_2.4.fa", [("rnd_rna", ["--rna"]), ("rnd_rna_pauto", ["--rna", "-p", "-1"]),
                                                                             ("rnd_rna_full_std", ["--rna", "--full-ref", "--dtw-std"])])):
        rng = np.random.default_rng(20241004 + k)
        reads = [(f"{kind}{i:03d}", float(rng.choice([2048.0, 8192.0])), float(rng.integers(-20, 40)), float(rng.uniform(700, 1500)), 4000.0,
                  signal(rng, int(rng.integers(1500, 5000)))) for i in range(40)]
        blow5 = os.path.join(rd, f"rnd_{kind}.blow5")
        write_blow5(blow5, reads, attrs=(("experiment_type", "rna" if kind == "rna" else "genomic_dna"), ("sequencing_kit", "unknown")), compress=True)
        for name, args in cases:
            out = subprocess.run([O.REF_DRIVER, "--model", os.path.join(GOLD, "models", f"syn{k}.f32"), "--kmer", str(k), *args,
                                  os.path.join(GOLD, "data", fasta), blow5], check=True, capture_output=True).stdout
            with open(os.path.join(rd, name + ".out"), "wb") as f:
                f.write(out)
            with open(os.path.join(rd, name + ".args"), "w") as f:
                f.write("\n".join([str(k), fasta, os.path.basename(blow5)] + args))
    print("random goldens written")


def random_goldens_long():
    """Six long synthetic DNA signals (30-50 k samples, 3-5 k events) through the reference's batch loop with -q beyond 2048
    events: end-to-end known answers for the row-strip path on both strands."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import write_blow5  # noqa: E402
    rd = os.path.join(GOLD, "random")
    os.makedirs(rd, exist_ok=True)
    rng = np.random.default_rng(20241005)

    def signal(n):
        dwell = rng.integers(4, 16, size=n // 4 + 2)
        levels = rng.normal(rng.uniform(400, 600), rng.uniform(40, 90), size=len(dwell))
        x = np.repeat(levels, dwell)[:n] + rng.normal(0, rng.uniform(1, 6), size=n)
        return np.clip(np.round(x), -2000, 4000).astype(np.int16)

    reads = [(f"dnalong{i:02d}", 8192.0, float(rng.integers(-20, 40)), float(rng.uniform(900, 1500)), 4000.0,
              signal(int(rng.integers(30000, 50000)))) for i in range(6)]
    blow5 = os.path.join(rd, "rnd_dnalong.blow5")
    write_blow5(blow5, reads, attrs=(("experiment_type", "genomic_dna"), ("sequencing_kit", "unknown")), compress=True)
    for name, args in (("rnd_dnalong_q3000", ["-q", "3000"]), ("rnd_dnalong_q2500_sam", ["-q", "2500", "--sam"])):
        out = subprocess.run([O.REF_DRIVER, "--model", os.path.join(GOLD, "models", "syn6.f32"), "--kmer", "6", *args,
                              os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), blow5], check=True, capture_output=True).stdout
        with open(os.path.join(rd, name + ".out"), "wb") as f:
            f.write(out)
        with open(os.path.join(rd, name + ".args"), "w") as f:
            f.write("\n".join(["6", "nCoV-2019.reference.fasta", os.path.basename(blow5)] + args))
    print("long random goldens written")


def main():
    O.build()
    assert os.path.exists(O.REF_DRIVER), "oracle/_ref missing (needs /root/reference)"
    for sub in ("data", "models", "cases"):
        os.makedirs(os.path.join(GOLD, sub), exist_ok=True)
    for f in ("nCoV-2019.reference.fasta", "rnasequin_sequences_2.4.fa", "sp1_dna.blow5", "sequin_rna.blow5",
              "sp1_dna.minimap2.paf", "sequin_rna.minimap2.paf"):
        dst = os.path.join(GOLD, "data", f)
        shutil.copyfile(os.path.join(REFTEST, f), dst)
        os.chmod(dst, 0o644)
    synth_levels(6, 1, 90, 12).tofile(os.path.join(GOLD, "models", "syn6.f32"))
    synth_levels(5, 2, 100, 14).tofile(os.path.join(GOLD, "models", "syn5.f32"))
    only = sys.argv[1:]  # case names: regenerate just these (the other fixtures are left as they are)
    if only == ["random_long"]:
        random_goldens_long()
        return
    for c in CASES:
        if not only or c[0] in only:
            run_case(*c)
    if only:
        return
    kernel_vectors()
    eval_goldens()
    random_goldens()
    random_goldens_long()


if __name__ == "__main__":
    main()
