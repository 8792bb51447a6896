"""Synthetic workloads at the alignment-stage boundary: seeded k-mer models, random references,
and reads drawn from the reference event arrays with over-segmentation and noise.  numpy only; used by bench.py
and the tests to produce inputs of the BASELINE.json shapes (there is no network for real datasets)."""
import numpy as np

from . import api


def kmer_levels(k, seed, mu=90.0, sd=12.0):
    """Seeded synthetic k-mer model: level_mean ~ N(mu, sd), lexicographic ACGT order."""
    rng = np.random.default_rng(seed)
    return rng.normal(mu, sd, size=4 ** k).astype(np.float32)


def random_sequence(n, seed):
    rng = np.random.default_rng(seed)
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, size=n)])


def _znorm_rows(q):
    """Row-wise z-normalisation with the reference's sequential fp32 recipe (src/sigfish.c:483-502)."""
    q = np.ascontiguousarray(q, np.float32)
    n = np.float32(q.shape[1])
    mean = (np.cumsum(q, axis=1, dtype=np.float32)[:, -1] / n).astype(np.float32)
    d = (q - mean[:, None]).astype(np.float32)
    var = (np.cumsum(d * d, axis=1, dtype=np.float32)[:, -1] / n).astype(np.float32)
    sd = np.sqrt(var).astype(np.float32)
    return ((q - mean[:, None]) / sd[:, None]).astype(np.float32)


def make_reads(ref: "api.RefModel", n_reads, qlen=250, seed=0, short_frac=0.05, noise=0.35, min_len=25):
    """Returns (queries float32[sum qlen], q_off int64[n+1], truth dict).

    Each read follows consecutive reference levels from a random (contig ∝ length, strand, start), every level
    repeated 1+Poisson(0.5) times (over-segmentation), plus N(0, noise²) in z-units, then z-normalised.
    `short_frac` of the reads are truncated to a length in [min_len, qlen) to exercise the ragged path.
    For RNA models (single strand) the event order is reversed, as the sequencer reads 3'->5'."""
    rng = np.random.default_rng(seed)
    rna = ref.reverse is None
    lens = ref.ref_lengths.astype(np.int64)
    contig = rng.choice(ref.num_ref, size=n_reads, p=lens / lens.sum())
    strand = np.zeros(n_reads, np.int8) if rna else rng.integers(0, 2, size=n_reads).astype(np.int8)
    # dwell -> k-mer index of each of the qlen events
    dwell = 1 + rng.poisson(0.5, size=(n_reads, qlen))
    cum = np.cumsum(dwell, axis=1)
    flags = np.zeros((n_reads, qlen + 1), np.int32)
    rows = np.repeat(np.arange(n_reads), qlen)
    cols = np.minimum(cum, qlen).ravel()
    flags[rows, cols] = 1  # cum is strictly increasing, so only the unused column qlen can repeat
    kidx = np.cumsum(flags[:, :qlen], axis=1)  # 0-based k-mer step of each event
    span = kidx[:, -1] + 1
    room = np.maximum(lens[contig] - span, 1)
    start = (rng.random(n_reads) * room).astype(np.int64)
    pos = np.minimum(start[:, None] + kidx, (lens[contig] - 1)[:, None])
    # gather levels from a flat copy of the arrays
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    flat_f = np.concatenate(ref.forward)
    flat = flat_f if rna else np.stack([flat_f, np.concatenate(ref.reverse)])
    base = offs[contig][:, None] + pos
    lv = flat[base] if rna else flat[strand[:, None].astype(np.int64), base]
    q = lv + rng.normal(0.0, noise, size=lv.shape).astype(np.float32)
    if rna:
        q = q[:, ::-1]
    q = _znorm_rows(q)
    # ragged tail: a fraction of reads keeps only their first L events (re-normalised)
    qlens = np.full(n_reads, qlen, np.int64)
    n_short = int(round(short_frac * n_reads)) if qlen > min_len else 0
    if n_short:
        which = rng.choice(n_reads, size=n_short, replace=False)
        qlens[which] = rng.integers(min_len, qlen, size=n_short)
    q_off = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    out = np.empty(int(q_off[-1]), np.float32)
    full = qlens == qlen
    # fast path: full-length rows are contiguous copies
    idx_full = np.nonzero(full)[0]
    if len(idx_full):
        dst = (q_off[idx_full][:, None] + np.arange(qlen)[None, :]).ravel()
        out[dst] = q[idx_full].ravel()
    for i in np.nonzero(~full)[0]:
        out[q_off[i]:q_off[i + 1]] = api.znormalise(q[i, :qlens[i]])
    truth = dict(contig=contig.astype(np.int32), strand=strand, start=start.astype(np.int64), span=span.astype(np.int64))
    return out, q_off, truth


def workload(name, n_reads=None, seed=0, golden_dir=None):
    """The BASELINE.json configurations as (RefModel, flag, queries, q_off, meta)."""
    import os
    gd = golden_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    import re
    m = re.fullmatch(r"ncov_r9_dna_q(\d+)", name)
    if m:  # q250 = configs[2]: synthetic R9 DNA reads x 30 kb nCoV reference, -q 250; other -q values for the long classes
        lv = kmer_levels(6, 1)
        qlen = int(m.group(1))
        ref = api.RefModel.from_fasta(os.path.join(gd, "data", "nCoV-2019.reference.fasta"), lv, 6, 0, qlen)
        flag, n = 0, n_reads or 100_000 * 250 // qlen
    elif name == "r10_dna_1mb_q250":  # configs[3]: synthetic R10 DNA reads x 1 Mb reference (k=9)
        lv = kmer_levels(9, 3)
        ref = api.RefModel.from_records([("synthetic_1Mb", random_sequence(1_000_000, 4))], lv, 9, 0, 250)
        flag, qlen, n = 0, 250, n_reads or 2_000
    elif name == "rna004_fullref_dtwstd_q250":  # configs[4]: RNA004 --full-ref --dtw-std x sequin transcriptome
        lv = kmer_levels(9, 3)
        flag = api.RNA | api.REF | api.DTW
        ref = api.RefModel.from_fasta(os.path.join(gd, "data", "rnasequin_sequences_2.4.fa"), lv, 9, flag, 250)
        qlen, n = 250, n_reads or 50_000
    elif name == "sequin_r9_rna_q250":  # configs[1] shape: R9 RNA x sequin 3'-slices
        lv = kmer_levels(5, 2, 100.0, 14.0)
        flag = api.RNA
        ref = api.RefModel.from_fasta(os.path.join(gd, "data", "rnasequin_sequences_2.4.fa"), lv, 5, flag, 250)
        qlen, n = 250, n_reads or 100_000
    else:
        raise ValueError(f"unknown workload {name}")
    q, q_off, truth = make_reads(ref, n, qlen=qlen, seed=seed)
    strands = 1 if ref.reverse is None else 2
    cells = int((q_off[1:] - q_off[:-1]).sum()) * int(ref.ref_lengths.sum()) * strands
    alg_bytes = int((4 * (q_off[1:] - q_off[:-1]) + 4 * strands * int(ref.ref_lengths.sum()) + 32).sum())
    meta = dict(name=name, n_reads=n, qlen=qlen, cells=cells, algorithmic_bytes=alg_bytes, truth=truth)
    return ref, flag, q, q_off, meta
