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


def test_host_units_under_sanitizers(tmp_path):
    """ASan + UBSan over the host-side C++ (reader, inflate, StreamVByte, events, query selection, planner): CPU build
    only -- the GPU pool runs no sanitizers."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "sigfish_amd", "csrc")
    exe = str(tmp_path / "host_asan")
    units = [os.path.join(csrc, u) for u in ("sfa_host.cpp", "host/blow5.cpp", "host/inflate.cpp", "host/events.cpp", "host/refio.cpp", "host/sam.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
                            "-I", csrc, "-o", exe, os.path.join(ROOT, "tests", "c", "host_asan.cpp"), *units, "-lz", "-lpthread"],
                           capture_output=True, timeout=600)
    if build.returncode != 0 and b"sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr.decode()[-2000:]
    data = os.path.join(ROOT, "tests", "golden")
    run = subprocess.run([exe, str(tmp_path), os.path.join(data, "data", "sp1_dna.blow5"), os.path.join(data, "data", "sequin_rna.blow5"),
                          os.path.join(data, "random", "rnd_dna.blow5")], capture_output=True, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr).decode()[-3000:]
    assert b"ERROR: AddressSanitizer" not in run.stderr and b"runtime error" not in run.stderr, run.stderr.decode()[-3000:]


def test_device_inflate_lane_decoder_on_the_host_with_sanitizers(tmp_path):
    """The per-lane DEFLATE decoder of the device-side BLOW5 reader (blow5_kernels.hpp: the body of blow5_inflate_kernel),
    compiled for the host: against zlib on 4 000 random streams (every block type, encoder setting and window size,
    arbitrary bytes behind the stream), too-small output slots, corrupted streams -- under ASan + UBSan, which the GPU pool
    cannot run."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "device_inflate_host")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                            os.path.join(ROOT, "tests", "c", "device_inflate_host.cpp"), "-lz"], capture_output=True, timeout=600)
    if build.returncode != 0 and b"sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr.decode()[-2000:]
    run = subprocess.run([exe, "4000", "3"], capture_output=True, timeout=900)
    assert run.returncode == 0 and b"4000 iterations, 0 failures" in run.stdout, (run.stdout + run.stderr).decode()[-3000:]


def test_strip_heights_cover_every_query_length(tmp_path):
    """sfa::strip_rows_per_lane (sdtw_strips.hpp), compiled for the host with hipcc: for every query length from 2049 to 400 000
    events the strips cover the query, the last one holds at least one row, the height is an instantiated one -- a violation
    would index rows outside the query on the device."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "strip_rows")
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "sigfish_amd", "csrc"), "-o", exe,
                            os.path.join(ROOT, "tests", "c", "strip_rows_host.hip")], capture_output=True, timeout=600)
    assert build.returncode == 0, build.stderr.decode()[-2000:]
    run = subprocess.run([exe], capture_output=True, timeout=300)
    assert run.returncode == 0 and run.stdout.startswith(b"0 violations"), (run.stdout + run.stderr).decode()
    assert b"0.800" in run.stdout  # at least 80 % of the rows of any query's strips are query rows
