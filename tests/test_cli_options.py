"""Switches of `sigfish dtw` that must not be silently swallowed (VERDICT r1 items 2, 3 / ADVICE): `--accel=no` names the
missing CPU path and fails, `--profile-cpu` takes yes/no only, `-q 0` is rejected at parse time.  All of these are decided
before any device is touched, so they run without a GPU."""
import os
import subprocess

from tests.util import ROOT, load_case

BIN = os.path.join(ROOT, "sigfish_amd", "bin", "sigfish-amd")


def _run(*extra):
    c = load_case("dna_default")
    return subprocess.run([BIN, "dtw", "--kmer-model", "/nonexistent.model", *extra, c["fasta"], c["blow5"]], capture_output=True, timeout=60)


def test_accel_no_fails_loudly():
    r = _run("--accel=no")
    assert r.returncode != 0 and r.stdout == b""
    err = r.stderr.decode()
    assert "--accel=no" in err and "no CPU alignment path" in err


def test_yes_no_switches_reject_other_values():
    for opt in ("--accel=maybe", "--profile-cpu=2"):
        r = _run(opt)
        assert r.returncode != 0 and "only accepts 'yes' or 'no'" in r.stderr.decode(), opt
    # accepted values get as far as the model file (which does not exist)
    for opt in ("--accel=yes", "--profile-cpu=yes", "--profile-cpu=no"):
        r = _run(opt)
        assert r.returncode != 0 and "nonexistent.model" in r.stderr.decode(), (opt, r.stderr.decode())


def test_query_size_zero_is_rejected_at_parse_time():
    r = _run("-q", "0")
    assert r.returncode != 0 and "Query size should larger than 0" in r.stderr.decode()


def test_help_lists_the_switches():
    r = subprocess.run([BIN, "dtw", "-h"], capture_output=True, timeout=60)
    assert r.returncode == 0
    for word in ("--profile-cpu=yes|no", "--accel=yes|no", "--sam", "--kmer-model"):
        assert word in r.stdout.decode(), word
