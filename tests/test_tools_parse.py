"""The helper scripts under tools/ (profiling, fuzzing, end-to-end runs) are not exercised by the CPU suite; at least
keep them syntactically alive, and check the pieces of them that need no GPU."""
import glob
import os
import py_compile
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile():
    for path in glob.glob(os.path.join(ROOT, "tools", "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        py_compile.compile(path, doraise=True)


def test_shell_tools_parse():
    for path in glob.glob(os.path.join(ROOT, "tools", "*.sh")):
        subprocess.run(["bash", "-n", path], check=True)


def test_blow5_writer_round_trips(tmp_path):
    """tools/make_blow5.py (every combination of record / signal compression) through the library's own reader."""
    import sigfish_amd as S
    src = os.path.join(ROOT, "tests", "golden", "data", "sp1_dna.blow5")
    want = list(S.Blow5File(src))
    for k, flags in enumerate(([], ["--compress"], ["--record-press", "zlib"], ["--signal-press", "svb-zd"])):
        dst = str(tmp_path / f"f{k}.blow5")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), src, dst, "--copies", "2", *flags], check=True,
                       capture_output=True)
        got = list(S.Blow5File(dst))
        assert len(got) == 2 * len(want)
        for j, (rid, meta, raw) in enumerate(got):
            wid, wmeta, wraw = want[j % len(want)]
            assert rid == f"{wid}_{j // len(want)}" and np.array_equal(raw, wraw)
            assert all(meta[f] == wmeta[f] for f in ("digitisation", "offset", "range", "sampling_rate"))
