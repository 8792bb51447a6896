#!/usr/bin/env python3
"""End-to-end wall of the command line: raw BLOW5 -> PAF through `sigfish-amd dtw` (SURVEY.md 8d "plus end-to-end wall").

    python tools/e2e_bench.py [--reads 400000] [--threads 16] [--ks 4096,512] [--ranks 2,4]

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
# a hung command line must not take the bench line with it (subprocess.run kills the child; its ranks follow: PR_SET_PDEATHSIG)
CLI_TIMEOUT_S = 300
MAKE_TIMEOUT_S = 600


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


def _run_cli(model, path, threads, k, extra=(), ranks=1, devices=None):
    """One timed run of the command line (process start to exit, output to a file); returns (seconds, output path owner's dir
    file, per-rank wall list | None)."""
    paf = os.path.join(os.path.dirname(path), "out.paf")
    cmd = [BIN, "dtw", "--kmer-model", model, "-t", str(threads), "-K", str(k), "-B", "2G", "--verbose", "3" if ranks > 1 else "0", *extra]
    if ranks > 1:
        cmd += ["--ranks", str(ranks)]
    if devices:
        cmd += ["--device", ",".join(str(d) for d in devices)]
    cmd += [os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), path]
    time.sleep(PAUSE_S)
    t0 = time.perf_counter()
    with open(paf, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, timeout=CLI_TIMEOUT_S)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd[1:])} failed: {r.stderr.decode()[-500:]}")
    walls = None
    if ranks > 1:
        import re
        walls = [float(x) for x in re.findall(r"\[dtw_main\] rank \d+/\d+ \([^)]*\): done after ([0-9.]+) sec", r.stderr.decode())]
        if len(walls) != ranks:
            raise RuntimeError(f"expected {ranks} rank lines on stderr, found {len(walls)}")
    return dt, paf, walls


def _check_rows(paf, copies, want_head, what):
    n, ok = 0, True
    with open(paf) as fi:
        for i, line in enumerate(fi):
            n += 1
            if i < 5:  # first copy of the fixture's reads: same rows as the fixture, read ids carry the copy suffix
                a, b = line.rstrip("\n").split("\t"), want_head[i].split("\t")
                ok = ok and a[0] == b[0] + "_0" and a[1:] == b[1:]
    if n != copies * 5 or not ok:
        raise RuntimeError(f"end-to-end output differs from the fixture on {what} ({n} rows)")
    return n


def _sha(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def measure(reads=400_000, threads=16, ks=(4096, 512), keep_dir=None, extra=(), long_file=True, ranks=(2,)):
    where = "given directory"
    d = keep_dir
    if d is None:
        d, where = _scratch_dir(reads * 10_000)  # ~9.6 KB per read uncompressed
    out = {"unit": "reads/s", "reads": 0, "host_threads": threads, "files_in": where,
           "page_cache": "warm: every generated file is synced and read once before its first timed run",
           "pause_before_each_run_s": PAUSE_S,
           "what": "raw BLOW5 -> PAF through `sigfish-amd dtw` (process start to exit), reference's DNA fixture replicated, nCoV reference",
           "ranks": "keys ending in _ranksG: the same command with --ranks G -- G processes on ONE GPU here, each mapping its byte "
                    "slice of the file with host_threads/G threads, output gathered in rank order and compared (sha256) with the "
                    "one-process output; rank_wall_s = seconds from the supervisor's start to each rank's exit"}
    try:
        lv = np.fromfile(os.path.join(GOLD, "models", "syn6.f32"), np.float32)
        model = os.path.join(d, "syn6.model")
        with open(model, "w") as f:
            f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
                f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
        copies = max(reads // 5, 1)
        want_head = open(os.path.join(GOLD, "cases", "dna_default.out")).read().splitlines()

        def sharded(path, k, key, n_copies, one_rank_sha):
            for g in ranks:
                dt, paf, walls = _run_cli(model, path, threads, k, extra, ranks=g)
                n = _check_rows(paf, n_copies, want_head, f"{key} with --ranks {g}")
                if _sha(paf) != one_rank_sha:
                    raise RuntimeError(f"--ranks {g} output differs from the one-process output on {key}")
                out[f"{key}_ranks{g}"] = round(n / dt, 1)
                out[f"{key}_ranks{g}_rank_wall_s"] = walls

        for kind, flags in (("uncompressed", []), ("compressed", ["--compress"])):
            path = os.path.join(d, kind + ".blow5")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", "sp1_dna.blow5"), path,
                            "--copies", str(copies), "--jobs", str(min(threads, 16)), *flags], check=True, capture_output=True, timeout=MAKE_TIMEOUT_S)
            out[kind + "_file_MB"] = round(os.path.getsize(path) / 1e6, 1)
            _warm(path)
            sha = None
            for k in ks:
                dt, paf, _ = _run_cli(model, path, threads, k, extra)
                n = _check_rows(paf, copies, want_head, f"the {kind} file at -K {k}")
                out["reads"] = n
                out[f"{kind}_K{k}"] = round(n / dt, 1)
                if k == ks[0]:
                    sha = _sha(paf)
            if kind == "compressed":
                sharded(path, ks[0], f"compressed_K{ks[0]}", copies, sha)
                # the same file with the records decompressed and parsed on the device (sfa_align_blow5)
                dt, paf, _ = _run_cli(model, path, threads, 8192, ["--gpu-parse", "--streams", "4"])
                n = _check_rows(paf, copies, want_head, "the compressed file with --gpu-parse")
                out["compressed_gpu_parse_K8192_streams4"] = round(n / dt, 1)
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
                            "--copies", str(long_copies), "--jobs", str(min(threads, 16)), "--compress"], check=True, capture_output=True, timeout=MAKE_TIMEOUT_S)
            _warm(path)
            dt, paf, _ = _run_cli(model, path, threads, 4096, extra)
            rows = _check_rows(paf, long_copies, want_head, "the long compressed file")
            out["long_file_reads"] = rows
            out["long_file_MB"] = round(os.path.getsize(path) / 1e6, 1)
            out["compressed_K4096_long_file"] = round(rows / dt, 1)
            sharded(path, 4096, "compressed_K4096_long_file", long_copies, _sha(paf))
            os.remove(path)
        out["parity"] = "row count and the first five rows (= the reference's PAF for the fixture) checked in every run; sharded runs byte-identical to the one-process run"
    finally:
        if not keep_dir:
            shutil.rmtree(d, ignore_errors=True)
    return out


def measure_sharded(ranks, devices, threads=16, reads=1_600_000):
    """The command line over SEVERAL GPUs of one node, one process per GPU (`--ranks G --device d0,d1,...`), against one process on
    the first of them, on ONE generated compressed file: whole-process reads/s of both, per-rank wall times, and whether the two
    outputs are the same bytes.  What bench.py adds to its line for --gpus N > 1.  Process start-up (0.3-0.4 s per rank, in
    parallel) is part of every figure: with `reads` fixed the speed-up is bounded by it, this is a functional check on real
    multi-GPU hardware first and a scaling figure second."""
    d, where = _scratch_dir(reads * 4200 * 2)
    out = {"unit": "reads/s", "ranks": ranks, "devices": list(devices), "host_threads": threads, "files_in": where,
           "what": f"raw compressed BLOW5 -> PAF through `sigfish-amd dtw --ranks {ranks} --device ...` (one process per GPU, each mapping its "
                   "byte slice of the file, output gathered in rank order) against one process on one GPU; process start to exit"}
    try:
        lv = np.fromfile(os.path.join(GOLD, "models", "syn6.f32"), np.float32)
        model = os.path.join(d, "syn6.model")
        with open(model, "w") as f:
            f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
                f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
        copies = max(reads // 5, 1)
        want_head = open(os.path.join(GOLD, "cases", "dna_default.out")).read().splitlines()
        path = os.path.join(d, "compressed.blow5")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", "sp1_dna.blow5"), path,
                        "--copies", str(copies), "--jobs", str(min(threads, 16)), "--compress"], check=True, capture_output=True, timeout=MAKE_TIMEOUT_S)
        out["file_MB"] = round(os.path.getsize(path) / 1e6, 1)
        _warm(path)
        dt, paf, _ = _run_cli(model, path, min(threads, 16), 4096, devices=[devices[0]])
        n = _check_rows(paf, copies, want_head, "the one-process run")
        sha = _sha(paf)
        out["reads"] = n
        out["one_process_one_gpu"] = round(n / dt, 1)
        dt, paf, walls = _run_cli(model, path, threads, 4096, ranks=ranks, devices=devices)
        n = _check_rows(paf, copies, want_head, f"the run with --ranks {ranks}")
        out[f"ranks{ranks}"] = round(n / dt, 1)
        out["rank_wall_s"] = walls
        out["identical_to_one_process"] = _sha(paf) == sha
        out["speedup"] = round(out[f"ranks{ranks}"] / out["one_process_one_gpu"], 2)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=400_000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--ks", default="4096,512")
    ap.add_argument("--no-long-file", action="store_true", help="skip the run on a file four times as long")
    ap.add_argument("--ranks", default="2", help="comma separated --ranks values for the sharded runs (empty: none)")
    ap.add_argument("--sharded", default=None, metavar="d0,d1,...", help="instead: measure_sharded() over these devices (one rank per entry)")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    if a.sharded:
        devs = [int(x) for x in a.sharded.split(",")]
        print(json.dumps(measure_sharded(len(devs), devs, a.threads, a.reads)))
        sys.exit(0)
    print(json.dumps(measure(a.reads, a.threads, tuple(int(k) for k in a.ks.split(",")), extra=a.extra, long_file=not a.no_long_file,
                             ranks=tuple(int(g) for g in a.ranks.split(",") if g))))
