"""A plain C99 program (tests/c/abi_host.c, gcc -std=c99 -pedantic) drives the C-ABI exactly as the reference's
align_db hook would; its rows must equal the oracle's.  Proves the header is C (not C++) and the boundary works
without Python in the loop."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.util import ROOT


def _build(tmp_path):
    exe = str(tmp_path / "abi_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_host.c"), "-L", os.path.join(ROOT, "sigfish_amd", "lib"),
                           "-lsigfish_amd", "-Wl,-rpath," + os.path.join(ROOT, "sigfish_amd", "lib"), "-o", exe])
    return exe


def test_header_is_strict_c99(tmp_path):
    _build(tmp_path)  # -Werror -pedantic


@pytest.mark.gpu
@pytest.mark.parametrize("rna", [False, True])
def test_c_host_matches_oracle(tmp_path, oracle, rna):
    import sigfish_amd as S
    exe = _build(tmp_path)
    rng = np.random.default_rng(4 + rna)
    lens = [333, 57, 1200]
    fw = [rng.normal(size=n).astype(np.float32) for n in lens]
    rv = None if rna else [rng.normal(size=n).astype(np.float32) for n in lens]
    offs = [0, 2, 5] if rna else [0, 0, 0]
    flag = S.RNA if rna else 0
    reads = []
    for i in range(13):
        ne = int(rng.integers(0, 420)) if i != 5 else 0
        a = min(50, ne)
        b = min(a + 250, ne)
        reads.append((ne, a, b, rng.normal(size=ne).astype(np.float32)))
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<3i", flag, len(lens), len(reads)))
        for i, n in enumerate(lens):
            f.write(struct.pack("<2i", n, offs[i]))
            f.write(fw[i].tobytes())
            if not rna:
                f.write(rv[i].tobytes())
        for ne, a, b, m in reads:
            f.write(struct.pack("<3q", ne, a, b))
            f.write(m.tobytes())
    subprocess.check_call([exe, fin, fout])
    got = np.fromfile(fout, S.RESULT_DTYPE)
    oref = oracle.RefSynth([f"c{i}" for i in range(3)], [n + 5 for n in lens], lens, offs, fw, rv)
    q = np.concatenate([m[a:b] for ne, a, b, m in reads])
    q_off = np.concatenate([[0], np.cumsum([b - a for ne, a, b, m in reads])]).astype(np.int64)
    want = oracle.align_batch(q, q_off, oref, flag, threads=4)
    assert np.array_equal(got["valid"], want["valid"])
    v = want["valid"] == 1
    assert got[v].tobytes() == want[v].tobytes()
