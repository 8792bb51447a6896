"""BLOW5 records decoded on the device (sfa_align_blow5, blow5_kernels.hpp): the DEFLATE decoder against zlib on every block
type, the StreamVByte / field parsing against the host reader, the whole entry point against sfa_align_raw on the
reference's fixtures written in all four compression combinations, and the fallback to the host reader."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

import sigfish_amd as S
from tests.util import GOLD, ROOT, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
    c = load_case("dna_default")
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    with S.Aligner(ref, c["flag"], device=0) as a:
        yield a


def _streams(rng):
    out = []
    for size in (0, 1, 2, 7, 100, 257, 258, 259, 1000, 5000, 32768, 70000):
        for kind in ("random", "text", "zeros", "deltas"):
            if kind == "random":
                b = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            elif kind == "text":
                b = (b"the quick brown fox jumps over the lazy dog " * (size // 44 + 1))[:size]
            elif kind == "zeros":
                b = bytes(size)
            else:  # what svb-zd signals look like: small values with structure
                b = (rng.integers(0, 6, size) * rng.integers(0, 40, size)).astype(np.uint8).tobytes()
            for level in (0, 1, 6, 9):  # level 0: stored blocks
                out.append((b, zlib.compress(b, level)))
            co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)  # fixed Huffman code
            out.append((b, co.compress(b) + co.flush()))
            co = zlib.compressobj(6, zlib.DEFLATED, 15, 1, zlib.Z_DEFAULT_STRATEGY)  # memLevel 1: many small dynamic blocks
            out.append((b, co.compress(b) + co.flush()))
            co = zlib.compressobj(1, zlib.DEFLATED, 9)  # 512-byte window
            out.append((b, co.compress(b) + co.flush()))
    return out


def test_device_inflate_equals_zlib(al):
    rng = np.random.default_rng(1)
    cases = _streams(rng)
    got = al.inflate_device([z for _, z in cases], cap_factor=1100, cap_extra=4096)  # zeros deflate ~1000x
    for (want, z), g in zip(cases, got):
        assert g == want, (len(want), len(z), None if g is None else len(g))


def test_device_inflate_rejects_what_zlib_rejects(al):
    rng = np.random.default_rng(2)
    base = zlib.compress(rng.integers(0, 50, 3000, dtype=np.uint8).tobytes(), 6)
    bad = []
    for _ in range(300):
        b = bytearray(base)
        k = int(rng.integers(0, 4))
        if k == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            del b[int(rng.integers(1, len(b))):]
        elif k == 2:
            b[-1] ^= 0xff  # Adler-32
        else:
            b = bytearray(rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8).tobytes())
        bad.append(bytes(b))
    got = al.inflate_device(bad, cap_factor=8, cap_extra=8192)
    for z, g in zip(bad, got):
        try:
            want = zlib.decompress(z)
        except zlib.error:
            want = None
        if want is None:
            assert g is None
        else:  # (a flipped bit may still be a valid stream with a matching checksum only by astronomical luck; trailing cuts cannot)
            assert g == want
    # an output slot that is too small is a refusal, not an overrun
    assert al.inflate_device([zlib.compress(bytes(100000), 6)], cap_factor=1, cap_extra=16) == [None]


def _records(path):
    """(record_zlib, signal_svb, [record bytes]) of a BLOW5 file, framing only"""
    b = open(path, "rb").read()
    rz, ss = b[9], b[14]
    (hl,) = struct.unpack_from("<I", b, 64)
    p = 68 + hl
    recs = []
    while b[p:p + 5] != b"5WOLB":
        (sz,) = struct.unpack_from("<Q", b, p)
        recs.append(b[p + 8:p + 8 + sz])
        p += 8 + sz
    return rz == 1, ss == 1, recs


@pytest.mark.parametrize("press", [[], ["--compress"], ["--record-press", "zlib"], ["--signal-press", "svb-zd"]])
@pytest.mark.parametrize("name", ["dna_default", "rna_default"])
def test_align_blow5_equals_align_raw(name, press, tmp_path):
    c = load_case(name)
    path = str(tmp_path / "x.blow5")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), c["blow5"], path, "--copies", "7", *press], check=True, capture_output=True)
    rz, ss, recs = _records(path)
    assert rz == ("--compress" in press or "zlib" in press) and ss == ("--compress" in press or "svb-zd" in press)
    reads = list(S.Blow5File(path))
    raw = np.concatenate([sig for _, _, sig in reads])
    raw_off = np.concatenate([[0], np.cumsum([len(sig) for _, _, sig in reads])]).astype(np.int64)
    meta = np.array([(m["digitisation"], m["offset"], m["range"]) for _, m, _ in reads])
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    with S.Aligner(ref, c["flag"], device=0) as al, S.Aligner(ref, c["flag"], devices=[0, 0]) as two:
        want_rows, want_info, want_ev = al.align_raw(raw, raw_off, meta, c["prefix_size"], c["query_size"], return_events=True)
        rec_off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.int64)
        blob = b"".join(recs)
        for a in (al, two):
            rows, info, heads, ev = a.align_blow5(blob, rec_off, rz, ss, c["prefix_size"], c["query_size"], return_events=True)
            assert rows.tobytes() == want_rows.tobytes() and info.tobytes() == want_info.tobytes() and ev.tobytes() == want_ev.tobytes()
            assert [h["read_id"] for h in heads] == [rid for rid, _, _ in reads]
            assert [h["n_samples"] for h in heads] == [len(sig) for _, _, sig in reads]
            assert [(h["digitisation"], h["offset"], h["range"]) for h in heads] == [tuple(m) for m in meta]
            assert [h["record_bytes"] for h in heads] == [len(r) for r in recs]
            p = a.profile()
            assert p["blow5_fallbacks"] == 0 and (a is two or p["decode_ms"] > 0), a._L.sfa_last_error()


def test_align_blow5_falls_back_to_the_host_reader(tmp_path):
    """A record the device decoder declines (here: a read id longer than its field row) sends the batch through the host
    reader: same rows.  A record nobody can read is an error."""
    c = load_case("dna_default")
    path = str(tmp_path / "x.blow5")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), c["blow5"], path, "--copies", "2", "--compress"], check=True, capture_output=True)
    rz, ss, recs = _records(path)
    ref = S.RefModel.from_fasta(c["fasta"], c["levels"], c["k"], c["flag"], c["query_size"])
    # re-wrap record 3 with a 125-character id (fits sfa_read_head_t, not the device's field row of 136 - ... bytes? it does:
    # use 127, the largest id the ABI holds; the device row holds 136) -> still on the device; 150 -> nobody holds it: error
    def rewrap(rec, new_id):
        p = zlib.decompress(rec)
        (il,) = struct.unpack_from("<H", p, 0)
        return zlib.compress(struct.pack("<H", len(new_id)) + new_id + p[2 + il:], 6)
    with S.Aligner(ref, c["flag"], device=0) as al:
        rec_off = lambda rs: np.concatenate([[0], np.cumsum([len(r) for r in rs])]).astype(np.int64)  # noqa: E731
        want = al.align_blow5(b"".join(recs), rec_off(recs), rz, ss)
        rs = list(recs)
        rs[3] = rewrap(recs[3], b"x" * 127)
        got = al.align_blow5(b"".join(rs), rec_off(rs), rz, ss)
        assert got[0].tobytes() == want[0].tobytes() and got[2][3]["read_id"] == "x" * 127
        # a slot overflow (record inflating to more than 4x + 4 KB): device declines, host reads it
        big = zlib.decompress(recs[5])
        rs = list(recs)
        rs[5] = zlib.compress(big + bytes(60000), 9)  # trailing auxiliary bytes that deflate to almost nothing
        before = al.profile()["blow5_fallbacks"]
        got = al.align_blow5(b"".join(rs), rec_off(rs), rz, ss)
        assert got[0].tobytes() == want[0].tobytes() and al.profile()["blow5_fallbacks"] == before + 1
        rs[5] = recs[5][:-7]  # truncated stream
        with pytest.raises(S.SfaError, match="record 5"):
            al.align_blow5(b"".join(rs), rec_off(rs), rz, ss)
