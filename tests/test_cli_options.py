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
    for word in ("--profile-cpu=yes|no", "--accel=yes|no", "--sam", "--kmer-model", "--ranks", "--shard", "--read-range", "--no-header", "--rank-buffer"):
        assert word in r.stdout.decode(), word


def test_sharding_switches_are_checked_before_any_device_is_touched():
    for extra, msg in ((["--ranks", "2", "--shard", "0/2"], "cannot be combined"), (["--shard", "3/2"], "0 <= r < G"),
                       (["--read-range", "7:3"], "A:B"), (["--shard", "0/2", "--read-range", "1:2"], "exclude each other")):
        r = _run(*extra)
        assert r.returncode != 0 and msg in r.stderr.decode(), (extra, r.stderr.decode())
    # a sharded run without a usable model: every rank fails on its own, the supervisor reports it and prints nothing
    r = _run("--ranks", "2")
    assert r.returncode != 0 and r.stdout == b"" and "rank of the sharded run failed" in r.stderr.decode()
