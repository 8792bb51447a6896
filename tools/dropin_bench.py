#!/usr/bin/env python3
"""What the drop-in buys a maintainer of the reference, in wall time, on one file (INTEGRATION.md quotes the numbers):

  reference      oracle/_ref/ref_driver      the reference's own batch loop + its CPU alignment (load_db -> process_db -> output_db,
                                             src/dtw_main.c:299-326), on a SAMPLE of the file (the whole file would take the better part
                                             of an hour on 8 threads), scaled by the read count -- per-read cost is constant by construction
  reference+hook oracle/_ref/ref_driver_acc  the same loop, unmodified, with oracle/ref_acc.patch: align_db() calls libsigfish_amd.so
                                             (what the patch ALONE gives: the reference's serial load / fork-join host stages stay);
                                             and with SIGFISH_ACC_RAW=1 (oracle/ref_acc_raw.patch): event detection + normalisation move
                                             to the device too (sfa_align_raw), parsing, loading and output stay the reference's
  sigfish-amd    sigfish_amd/bin/sigfish-amd dtw   this repo's command line at the same -K / -t, and at its own defaults

All three at the reference's defaults -K 512 -t 8 (src/sigfish.c:1124-1128); the reference-side runs with --profile-cpu, i.e. its
stage timers (src/sigfish.c:1021-1040).  Outputs must be byte-identical (the sample's rows = the first rows of the others).
Run on the GPU box:  python tools/dropin_bench.py [--reads 400000] [--sample 4000] > profiles/r04_dropin.json"""
import argparse
import hashlib
import itertools
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "sigfish_amd", "bin", "sigfish-amd")
REF = os.path.join(ROOT, "oracle", "_ref")


def sha(b):
    return hashlib.sha256(b).hexdigest()[:16]


def run(cmd, out_path, env=None):
    time.sleep(1.0)  # the previous process's device contexts are gone (tools/e2e_bench.py: PAUSE_S)
    t0 = time.perf_counter()
    with open(out_path, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed ({r.returncode}): {r.stderr.decode()[-400:]}")
    return dt, r.stderr.decode()


def stage_timers(err):
    m = re.search(r"parse ([0-9.]+) events ([0-9.]+) normalise ([0-9.]+) dtw ([0-9.]+)", err)
    return None if not m else dict(zip(("parse_s", "events_s", "normalise_s", "dtw_s"), map(float, m.groups())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=400_000)
    ap.add_argument("--sample", type=int, default=4000, help="reads of the file the unpatched reference is timed on")
    ap.add_argument("-K", type=int, default=512)
    ap.add_argument("-t", type=int, default=8)
    a = ap.parse_args()
    for exe in ("ref_driver", "ref_driver_acc"):
        if not os.path.exists(os.path.join(REF, exe)):
            raise SystemExit(f"oracle/_ref/{exe} is missing: `python -c 'import __graft_entry__ as g; g.build()'` where /root/reference is mounted")
    d = tempfile.mkdtemp(prefix="sfa_dropin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"file_reads": a.reads, "K": a.K, "t": a.t, "what": __doc__.split("\n\n")[0]}
    try:
        lv = np.fromfile(os.path.join(GOLD, "models", "syn6.f32"), np.float32)
        model = os.path.join(d, "syn6.model")
        with open(model, "w") as f:
            f.write("#k\t6\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for kmer, v in zip(itertools.product("ACGT", repeat=6), lv):
                f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
        fasta = os.path.join(GOLD, "data", "nCoV-2019.reference.fasta")
        files = {}
        for name, n in (("full", a.reads), ("sample", a.sample)):
            files[name] = os.path.join(d, name + ".blow5")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", "sp1_dna.blow5"), files[name],
                            "--copies", str(max(n // 5, 1)), "--jobs", "16", "--compress"], check=True, capture_output=True)
            os.sync()
            open(files[name], "rb").read()
        ref_args = ["--model", os.path.join(GOLD, "models", "syn6.f32"), "--kmer", "6", "-t", str(a.t), "-K", str(a.K), "--profile-cpu"]
        # 1. the reference as it is, on the sample
        dt, err = run([os.path.join(REF, "ref_driver"), *ref_args, fasta, files["sample"]], os.path.join(d, "ref.paf"))
        ref_rows = open(os.path.join(d, "ref.paf"), "rb").read()
        n_sample = ref_rows.count(b"\n")
        out["reference_cpu"] = {"reads": n_sample, "wall_s": round(dt, 3), "stages": stage_timers(err),
                                "reads_per_s": round(n_sample / dt, 1), "wall_s_scaled_to_file": round(dt * a.reads / n_sample, 1)}
        # 2. the reference + the hook, whole file
        dt, err = run([os.path.join(REF, "ref_driver_acc"), *ref_args, fasta, files["full"]], os.path.join(d, "acc.paf"))
        acc_rows = open(os.path.join(d, "acc.paf"), "rb").read()
        n = acc_rows.count(b"\n")
        out["reference_with_hook"] = {"reads": n, "wall_s": round(dt, 3), "stages": stage_timers(err), "reads_per_s": round(n / dt, 1)}
        # 2b. ... with oracle/ref_acc_raw.patch switched on: event detection and normalisation on the device as well (sfa_align_raw)
        dt, err = run([os.path.join(REF, "ref_driver_acc"), *ref_args, fasta, files["full"]], os.path.join(d, "accraw.paf"),
                      env=dict(os.environ, SIGFISH_ACC_RAW="1"))
        raw_rows = open(os.path.join(d, "accraw.paf"), "rb").read()
        out["reference_with_raw_hook"] = {"reads": raw_rows.count(b"\n"), "wall_s": round(dt, 3), "stages": stage_timers(err),
                                          "reads_per_s": round(raw_rows.count(b"\n") / dt, 1), "identical_to_reference_with_hook": raw_rows == acc_rows,
                                          "note": "stages: `dtw_s` covers events + normalise + align (one call into the library)"}
        # 3. this repo's command line: same -K / -t, then its defaults
        for key, extra in (("sigfish_amd_same_K_t", ["-K", str(a.K), "-t", str(a.t)]), ("sigfish_amd_defaults_t16", ["-t", "16"]),
                           ("sigfish_amd_same_K_t_profile_cpu", ["-K", str(a.K), "-t", str(a.t), "--profile-cpu=yes"])):
            dt, err = run([BIN, "dtw", "--kmer-model", model, "-B", "2G", "--verbose", "3" if "profile" in key else "0", *extra, fasta, files["full"]],
                          os.path.join(d, "sfa.paf"))
            rows = open(os.path.join(d, "sfa.paf"), "rb").read()
            rec = {"reads": rows.count(b"\n"), "wall_s": round(dt, 3), "reads_per_s": round(rows.count(b"\n") / dt, 1),
                   "identical_to_reference_with_hook": rows == acc_rows}
            if "profile" in key:
                rec["stages"] = {k.lower() + "_s": float(v) for k, v in re.findall(r"- (Parse|Events|Normalise|DTW) time: ([0-9.]+) sec", err)}
            out[key] = rec
        # the copies' read ids differ by their suffix only; rows of the sample = the first rows of the full runs (same reads, same order)
        out["reference_rows_equal_first_rows_with_hook"] = acc_rows[:len(ref_rows)] == ref_rows
        out["sha256_16"] = {"reference_sample": sha(ref_rows), "with_hook": sha(acc_rows)}
        out["speedup_hook_over_reference"] = round(out["reference_cpu"]["wall_s_scaled_to_file"] / out["reference_with_hook"]["wall_s"], 1)
        out["speedup_cli_over_hook_same_K_t"] = round(out["reference_with_hook"]["wall_s"] / out["sigfish_amd_same_K_t"]["wall_s"], 2)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
