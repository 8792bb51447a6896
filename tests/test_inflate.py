"""The BLOW5 reader's own DEFLATE decoder (sigfish_amd/csrc/host/inflate.cpp) against zlib: every block type, window
sizes, strategies, and corrupt input (which it must decline, not crash on -- the reader then falls back to zlib)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from sigfish_amd import _lib


def _inflate(b, cap=None):
    L = _lib.load()
    cap = cap or max(len(b) * 1100, 1 << 16)
    out = np.zeros(cap, np.uint8)
    r = L.sfa_inflate_zlib(bytes(b), len(b), out.ctypes.data_as(C.c_void_p), cap)
    return r, out[:max(r, 0)].tobytes()


def _payloads():
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 5, 50, 300, 4000, 70000):
        yield rng.integers(0, 256, n, dtype=np.uint8).tobytes()                      # incompressible -> stored / literals
        yield rng.integers(0, 4, n, dtype=np.uint8).tobytes()                        # low entropy
        yield (b"abcabcabd" * (n // 9 + 1))[:n]                                      # short distances, overlapping copies
        yield bytes(n)                                                               # distance 1, longest matches
        yield np.cumsum(rng.integers(-3, 4, n)).astype(np.int16).tobytes()[:n]       # signal-like


@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
def test_matches_zlib(strategy):
    for data in _payloads():
        for level in (0, 1, 6, 9):
            for wbits in (15, 9):
                co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
                comp = co.compress(data) + co.flush()
                r, out = _inflate(comp)
                assert r == len(data) and out == data, (len(data), level, wbits, strategy, r)


def test_output_capacity_and_declines():
    data = bytes(range(256)) * 40
    comp = zlib.compress(data)
    assert _inflate(comp, cap=100)[0] == -4            # SFA_ERANGE
    assert _inflate(comp[:2])[0] == -1                 # too short
    assert _inflate(b"\x78\xbb" + comp[2:])[0] == -1   # FDICT set: preset dictionaries are left to zlib
    assert _inflate(b"\x1f\x8b" + comp[2:])[0] == -1   # gzip magic, not a zlib stream
    bad = bytearray(comp)
    bad[-1] ^= 0x55                                    # Adler-32 trailer
    assert _inflate(bad)[0] == -1


def test_corrupt_streams_never_crash():
    rng = np.random.default_rng(1)
    decoded = 0
    for it in range(3000):
        n = int(rng.integers(1, 5000))
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes() if it % 2 else (b"xyz" * n)[:n]
        comp = bytearray(zlib.compress(data, int(rng.integers(0, 10))))
        if it % 3 == 0:
            comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
        elif it % 3 == 1:
            comp = comp[:int(rng.integers(0, len(comp)))]
        else:
            comp += bytes(rng.integers(0, 256, 7, dtype=np.uint8))  # trailing bytes are ignored, like zlib does
        r, out = _inflate(comp, 1 << 18)
        if r >= 0 and it % 3:
            assert out == data
            decoded += 1
    assert decoded > 900


def _inflate_pair(b0, b1, cap=1 << 18):
    L = _lib.load()
    o0, o1 = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
    lens = np.zeros(2, np.int64)
    rc = L.sfa_inflate_zlib_pair(bytes(b0), len(b0), bytes(b1), len(b1), o0.ctypes.data_as(C.c_void_p), cap, o1.ctypes.data_as(C.c_void_p), cap,
                                 lens.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc == 0
    return (int(lens[0]), o0[:max(int(lens[0]), 0)].tobytes()), (int(lens[1]), o1[:max(int(lens[1]), 0)].tobytes())


def test_two_streams_side_by_side_give_what_each_gives_alone():
    """The reader's threads inflate records two at a time (the streams' symbol loops take turns): every pairing of the payload
    kinds, block types and lengths -- short next to long, stored next to dynamic -- must give each stream's own bytes."""
    rng = np.random.default_rng(3)
    comps = []
    for data in _payloads():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (1, zlib.Z_RLE)):
            co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
            comps.append((co.compress(data) + co.flush(), data))
    order = rng.permutation(len(comps))
    for i in range(0, len(order) - 1):
        (c0, d0), (c1, d1) = comps[order[i]], comps[order[i + 1]]
        (r0, o0), (r1, o1) = _inflate_pair(c0, c1, cap=1 << 17)
        assert (r0, o0) == (len(d0), d0) and (r1, o1) == (len(d1), d1), (i, len(d0), len(d1))


def test_a_corrupt_stream_does_not_disturb_its_partner():
    rng = np.random.default_rng(4)
    good = 0
    for it in range(2000):
        n0, n1 = int(rng.integers(1, 5000)), int(rng.integers(1, 5000))
        d0 = rng.integers(0, 256, n0, dtype=np.uint8).tobytes() if it % 2 else (b"xyz" * n0)[:n0]
        d1 = np.cumsum(rng.integers(-3, 4, n1)).astype(np.int16).tobytes()[:n1]
        c0, c1 = bytearray(zlib.compress(d0, int(rng.integers(0, 10)))), zlib.compress(d1, 6)
        kind = it % 4
        if kind == 0:
            c0[int(rng.integers(0, len(c0)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            c0 = c0[:int(rng.integers(0, len(c0)))]
        elif kind == 2:
            c0 += bytes(rng.integers(0, 256, 7, dtype=np.uint8))
        first_bad = it % 3 == 0  # the damaged stream first or second
        a, b = _inflate_pair(c0, c1) if first_bad else _inflate_pair(c1, c0)[::-1]
        assert b == (len(d1), d1)            # the intact partner always decodes
        alone = _inflate(c0, 1 << 18)
        assert a == alone                    # ... and the damaged one gets exactly what it gets alone
        good += a[0] >= 0
    assert good > 900
    # output too small for one of them, null arguments
    (r0, _), (r1, o1) = _inflate_pair(zlib.compress(bytes(100000)), zlib.compress(b"abc"), cap=1000)
    assert r0 == -4 and (r1, o1) == (3, b"abc")
