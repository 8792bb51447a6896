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
