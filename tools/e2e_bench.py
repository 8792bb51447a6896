#!/usr/bin/env python3
"""End-to-end wall of the command line: raw BLOW5 -> PAF through `sigfish-amd dtw` (SURVEY.md 8d "plus end-to-end wall").

    python tools/e2e_bench.py [--reads 400000] [--threads 16] [--ks 4096,512]

Generates two files with tools/make_blow5.py from the reference's own DNA fixture replicated --reads/5 times --
uncompressed, and zlib records + svb-zd signals (what real files look like) -- runs the command line on each at every -K
(whole process: start-up, reference upload, file mapping, host stages, GPU stages, output) and prints one JSON object.
bench.py calls measure() for the `end_to_end` key of its line (never `value`)."""
import argparse
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sigfish_amd", "bin", "sigfish-amd")
GOLD = os.path.join(ROOT, "tests", "golden")


# Every timed run is its own process; the driver is still tearing down the previous one's device contexts for a moment after it
# exits, and a process started right then waits for that inside its own start-up (measured: 2 contexts in 0.20-0.26 s back to
# back, 0.11-0.14 s after a second's pause; profiles/r03_logs/rejected_cli_async_init_run_ahead_and_fast_exit.log).  The pause is
# not timed: the figure is the wall time of ONE command, as a user would run it.
PAUSE_S = 1.0


def _scratch_dir(need_bytes):
    """Where the generated files go: memory-backed /dev/shm when it has room (the leg measures the pipeline, not the box's
    disk), else the ordinary temporary directory."""
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 2 * need_bytes:
            return tempfile.mkdtemp(prefix="sfa_e2e_", dir="/dev/shm"), "/dev/shm (memory-backed)"
    except OSError:
        pass
    return tempfile.mkdtemp(prefix="sfa_e2e_"), tempfile.gettempdir()


def _warm(path):
    """The file was written seconds ago: flush it, then read it once, so that no timed run pays the write-back of dirty pages or
    a cold page cache (round 2's driver run recorded 0.15 M reads/s for the FIRST configuration on a 3.8 GB file and 0.39-0.49 M
    for every later one: that first figure was the box's disk, not the pipeline)."""
    os.sync()
    with open(path, "rb", buffering=0) as f:
        while f.read(1 << 24):
            pass


def measure(reads=400_000, threads=16, ks=(4096, 512), keep_dir=None, extra=(), long_file=True):
    where = "given directory"
    d = keep_dir
    if d is None:
        d, where = _scratch_dir(reads * 10_000)  # ~9.6 KB per read uncompressed
    out = {"unit": "reads/s", "reads": 0, "host_threads": threads, "files_in": where,
           "page_cache": "warm: every generated file is synced and read once before its first timed run",
           "pause_before_each_run_s": PAUSE_S,
           "what": "raw BLOW5 -> PAF through `sigfish-amd dtw` (process start to exit), reference's DNA fixture replicated, nCoV reference"}
    try:
        lv = np.fromfile(os.path.join(GOLD, "models", "syn6.f32"), np.float32)
        model = os.path.join(d, "syn6.model")
        with open(model, "w") as f:
            f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
                f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
        copies = max(reads // 5, 1)
        want_head = open(os.path.join(GOLD, "cases", "dna_default.out")).read().splitlines()
        for kind, flags in (("uncompressed", []), ("compressed", ["--compress"])):
            path = os.path.join(d, kind + ".blow5")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", "sp1_dna.blow5"), path,
                            "--copies", str(copies), "--jobs", str(min(threads, 16)), *flags], check=True, capture_output=True)
            out[kind + "_file_MB"] = round(os.path.getsize(path) / 1e6, 1)
            _warm(path)
            for k in ks:
                paf = os.path.join(d, "out.paf")
                time.sleep(PAUSE_S)
                t0 = time.perf_counter()
                with open(paf, "wb") as fo:
                    r = subprocess.run([BIN, "dtw", "--kmer-model", model, "-t", str(threads), "-K", str(k), "-B", "2G", "--verbose", "0", *extra,
                                        os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), path], stdout=fo, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(f"sigfish-amd dtw failed on the {kind} file at -K {k}: {r.stderr.decode()[-500:]}")
                n = 0
                ok = True
                with open(paf) as fi:
                    for i, line in enumerate(fi):
                        n += 1
                        if i < 5:  # first copy of the fixture's reads: same rows as the fixture, read ids carry the copy suffix
                            a, b = line.rstrip("\n").split("\t"), want_head[i].split("\t")
                            ok = ok and a[0] == b[0] + "_0" and a[1:] == b[1:]
                if n != copies * 5 or not ok:
                    raise RuntimeError(f"end-to-end output differs from the fixture on the {kind} file at -K {k} ({n} rows)")
                out["reads"] = n
                out[f"{kind}_K{k}"] = round(n / dt, 1)
            if kind == "compressed":  # the same file with the records decompressed and parsed on the device (sfa_align_blow5)
                time.sleep(PAUSE_S)
                t0 = time.perf_counter()
                with open(os.path.join(d, "out.paf"), "wb") as fo:
                    r = subprocess.run([BIN, "dtw", "--kmer-model", model, "-t", str(threads), "-K", "8192", "-B", "2G", "--verbose", "0",
                                        "--gpu-parse", "--streams", "4", os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), path],
                                       stdout=fo, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                rows = sum(1 for _ in open(os.path.join(d, "out.paf")))
                if r.returncode != 0 or rows != copies * 5:
                    raise RuntimeError(f"sigfish-amd dtw --gpu-parse failed on the compressed file: {r.stderr.decode()[-300:]}")
                out["compressed_gpu_parse_K8192_streams4"] = round(rows / dt, 1)
            os.remove(path)
        # ... and a file four times as long (real files hold millions of reads): what a process costs besides its reads -- HIP
        # runtime, code objects, contexts, exit: 0.3-0.4 s -- is 40 % of a 400 000-read run and 15 % of this one
        long_copies = 4 * copies
        try:
            st = os.statvfs(d)
            room = st.f_bavail * st.f_frsize
        except OSError:
            room = 0
        if long_file and where.startswith("/dev/shm") and room > 3 * long_copies * 5 * 4200:  # (memory-backed scratch only: 6.6 GB at the default size)
            path = os.path.join(d, "compressed_long.blow5")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", "sp1_dna.blow5"), path,
                            "--copies", str(long_copies), "--jobs", str(min(threads, 16)), "--compress"], check=True, capture_output=True)
            _warm(path)
            time.sleep(PAUSE_S)
            t0 = time.perf_counter()
            with open(os.path.join(d, "out.paf"), "wb") as fo:
                r = subprocess.run([BIN, "dtw", "--kmer-model", model, "-t", str(threads), "-K", "4096", "-B", "2G", "--verbose", "0", *extra,
                                    os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), path], stdout=fo, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            rows = sum(1 for _ in open(os.path.join(d, "out.paf")))
            if r.returncode != 0 or rows != long_copies * 5:
                raise RuntimeError(f"sigfish-amd dtw failed on the long compressed file: {r.stderr.decode()[-300:]}")
            out["long_file_reads"] = rows
            out["long_file_MB"] = round(os.path.getsize(path) / 1e6, 1)
            out["compressed_K4096_long_file"] = round(rows / dt, 1)
            os.remove(path)
        out["parity"] = "row count and the first five rows (= the reference's PAF for the fixture) checked in every run"
    finally:
        if not keep_dir:
            shutil.rmtree(d, ignore_errors=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=400_000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--ks", default="4096,512")
    ap.add_argument("--no-long-file", action="store_true", help="skip the run on a file four times as long")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    print(json.dumps(measure(a.reads, a.threads, tuple(int(k) for k in a.ks.split(",")), extra=a.extra, long_file=not a.no_long_file)))
