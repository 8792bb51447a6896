"""End to end: the `sigfish-amd dtw` command line (own BLOW5/FASTA/model readers, event detection, query window,
GPU alignment through the C-ABI, PAF writer) must print exactly what the compiled reference printed for the
same files and flags (tests/golden/cases/*.out) -- "PAF identical to CPU on test/test.sh"."""
import glob
import itertools
import os
import subprocess

import numpy as np
import pytest

from tests.util import GOLD, ROOT, case_names, load_case

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "sigfish_amd", "bin", "sigfish-amd")


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    d = tmp_path_factory.mktemp("models")
    out = {}
    for k in (5, 6):
        lv = np.fromfile(os.path.join(GOLD, "models", f"syn{k}.f32"), np.float32)
        p = d / f"syn{k}.model"
        with open(p, "w") as f:
            f.write(f"#k\t{k}\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for kmer, v in zip(itertools.product("ACGT", repeat=k), lv):
                f.write("%s\t%.4f\t1.5000\t1.0\t1.0\n" % ("".join(kmer), v))
        out[k] = str(p)
    return out


@pytest.mark.parametrize("name", case_names())
def test_cli_matches_reference_output(name, models):
    assert os.path.exists(BIN), "build with `make -C sigfish_amd/csrc`"
    c = load_case(name)
    args = [str(a) for a in c["args"]]
    cmd = [BIN, "dtw", "--kmer-model", models[c["k"]], "--verbose", "0", *args, c["fasta"], c["blow5"]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]


@pytest.fixture(scope="module")
def text_files(tmp_path_factory):
    """the two fixtures of the reference as SLOW5 ASCII (same reads, same ids; the RNA one with two auxiliary columns)"""
    import sys
    d = tmp_path_factory.mktemp("slow5")
    out = {}
    for sub, name, aux in (("data", "sp1_dna", 0), ("data", "sequin_rna", 2), ("random", "rnd_dna", 1), ("random", "rnd_rna", 0), ("random", "rnd_dnalong", 0)):
        out[name + ".blow5"] = str(d / (name + ".slow5"))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, sub, name + ".blow5"), out[name + ".blow5"],
                        "--copies", "1", "--ascii", "--keep-ids", "--aux", str(aux)], check=True, capture_output=True)
    return out


@pytest.mark.parametrize("extra", [[], ["--host-events"], ["--ranks", "2"], ["--ranks", "3", "-K", "2"]])
@pytest.mark.parametrize("name", case_names())
def test_cli_reads_slow5_ascii(name, extra, models, text_files):
    """The text twin of the format (slow5_open takes either): the same reads from a .slow5 file print the reference's output,
    one process or read-sharded (a rank's part of a text file = the lines starting in its byte slice)."""
    c = load_case(name)
    if os.path.basename(c["blow5"]) not in text_files:
        pytest.skip("case on another input file")
    args = [str(a) for a in c["args"]]
    cmd = [BIN, "dtw", "--kmer-model", models[c["k"]], "--verbose", "0", *extra, *args, c["fasta"], text_files[os.path.basename(c["blow5"])]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]


@pytest.mark.parametrize("extra", [[], ["--ranks", "3", "-K", "7"]])
@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLD, "random", "*.args"))))
def test_cli_reads_slow5_ascii_random_goldens(name, extra, models, text_files):
    """the 40-read random-signal files (negative samples, ragged lengths) as text"""
    k, fasta, blow5, *args = open(os.path.join(GOLD, "random", name + ".args")).read().split("\n")
    cmd = [BIN, "dtw", "--kmer-model", models[int(k)], "--verbose", "0", *args, *extra, os.path.join(GOLD, "data", fasta), text_files[blow5]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == open(os.path.join(GOLD, "random", name + ".out")).read()


def test_cli_slow5_ascii_ranges_and_refusals(models, text_files):
    c = load_case("dna_default")
    base = [BIN, "dtw", "--kmer-model", models[6], "--verbose", "0"]
    files = [c["fasta"], text_files["sp1_dna.blow5"]]
    parts = [subprocess.run([*base, "--read-range", rg, *files], capture_output=True, timeout=300) for rg in ("0:2", "2:3", "3:")]
    assert all(p.returncode == 0 for p in parts) and "".join(p.stdout.decode() for p in parts) == c["out_text"]
    bad = subprocess.run([*base, "--gpu-parse", *files], capture_output=True, timeout=120)
    assert bad.returncode != 0 and "parsed on the host threads" in bad.stderr.decode()
    cut = files[1] + ".cut"
    open(cut, "w").write(open(files[1]).read()[:-1])  # the last newline gone
    bad = subprocess.run([*base, c["fasta"], cut], capture_output=True, timeout=120)
    assert bad.returncode != 0 and "newline" in bad.stderr.decode()


@pytest.mark.parametrize("stage", ["--host-events", "--gpu-parse"])
@pytest.mark.parametrize("name", ["dna_default", "dna_from_end", "rna_default", "rna_full_ref_dtw_std", "dna_sam", "rna_sam", "rna_q2000_sam", "rna_q2500", "rna_q4200_full_sam"])
def test_cli_host_events_path(name, stage, models):
    """--host-events forces event detection onto host threads, --gpu-parse moves the record decompression / parsing onto the
    GPU as well (the default up to two devices: events on the GPU, records on host threads); output must be identical."""
    c = load_case(name)
    args = [str(a) for a in c["args"]]
    cmd = [BIN, "dtw", "--kmer-model", models[c["k"]], "--verbose", "0", stage, *args, c["fasta"], c["blow5"]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("extra", [[], ["--host-events"], ["--sam"]])
def test_cli_small_batches_keep_order(models, k, extra):
    """-K 1..3 forces up to eight batches through the four-slot / two-context pipeline; output order and content
    must not change (8 reads: batches overlap on the device and in the output thread)."""
    c = load_case("rna_sam" if "--sam" in extra else "rna_default")
    cmd = [BIN, "dtw", "--kmer-model", models[5], "--verbose", "0", "-K", str(k), "--rna", *extra, c["fasta"], c["blow5"]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]


def _device_args():
    from tests.util import distinct_device_list
    lists = [["--device", "0,0"], ["--device", "0,0", "--streams", "1"], ["--streams", "3"]]
    devs = distinct_device_list()
    if devs:  # a box with several GPUs: every one of them, without a code change
        lists += [["--device", ",".join(map(str, devs))], ["--device", ",".join(map(str, devs)), "--streams", "1"]]
    return lists


@pytest.mark.parametrize("extra", _device_args(), ids=lambda e: "_".join(e).replace("--", "").replace(",", ""))
def test_cli_device_list_and_streams(models, extra):
    """Batches dealt to several contexts / devices in turn (one GPU listed twice; every GPU of a multi-GPU box) come out in order."""
    c = load_case("rna_default")
    cmd = [BIN, "dtw", "--kmer-model", models[5], "--verbose", "0", "-K", "1", "--rna", *extra, c["fasta"], c["blow5"]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]
    bad = subprocess.run([BIN, "dtw", "--kmer-model", models[5], "--device", "0;1", c["fasta"], c["blow5"]], capture_output=True)
    assert bad.returncode != 0 and "comma separated" in bad.stderr.decode()


def test_cli_empty_file_and_odd_batch_settings(models, tmp_path):
    import sys
    c = load_case("dna_default")
    empty = str(tmp_path / "empty.blow5")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "make_blow5.py"), c["blow5"], empty, "--copies", "0"], check=True, capture_output=True)
    for extra in ([], ["--host-events"]):
        r = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "0", *extra, c["fasta"], empty], capture_output=True, timeout=120)
        assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()
    for extra in (["-K", "100000"], ["-B", "1K"]):   # batch larger than the file; byte cap of one record
        r = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "0", *extra, c["fasta"], c["blow5"]], capture_output=True, timeout=120)
        assert r.returncode == 0 and r.stdout.decode() == c["out_text"], r.stderr.decode()


RANDOM_CASES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLD, "random", "*.args")))


@pytest.mark.parametrize("name", RANDOM_CASES)
@pytest.mark.parametrize("extra", [[], ["-K", "7"], ["--host-events"], ["--gpu-parse"], ["--gpu-parse", "-K", "7"], ["--host-parse", "-K", "7"]])
def test_cli_random_signal_goldens(name, extra, models):
    """Synthetic step signals in compressed BLOW5 files (40 reads each) whose PAF / SAM text the compiled reference
    printed (tests/golden/random, oracle/make_golden.py): the command line must print the same, through device-side and
    host-side event detection, in one batch and in several, with the records parsed on host threads (--host-parse) or on the device
    (--gpu-parse)."""
    k, fasta, blow5, *args = open(os.path.join(GOLD, "random", name + ".args")).read().split("\n")
    cmd = [BIN, "dtw", "--kmer-model", models[int(k)], "--verbose", "0", *args, *extra, os.path.join(GOLD, "data", fasta),
           os.path.join(GOLD, "random", blow5)]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == open(os.path.join(GOLD, "random", name + ".out")).read()


@pytest.mark.parametrize("ranks", [2, 3])
@pytest.mark.parametrize("name", case_names())
def test_cli_ranks_print_the_single_process_output(name, ranks, models):
    """`--ranks G`: the whole pipeline sharded by reads over G processes (here all on device 0), each mapping its byte slice of
    the file, their output gathered in rank order: byte-identical to what the compiled reference printed, PAF and SAM (header
    once), with fewer reads than ranks in some cases (empty shards)."""
    c = load_case(name)
    args = [str(a) for a in c["args"]]
    cmd = [BIN, "dtw", "--kmer-model", models[c["k"]], "--verbose", "0", "--ranks", str(ranks), *args, c["fasta"], c["blow5"]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]


@pytest.mark.parametrize("name", RANDOM_CASES)
@pytest.mark.parametrize("extra", [["--ranks", "2"], ["--ranks", "3", "-K", "7"], ["--ranks", "4", "-K", "3", "--gpu-parse"],
                                   ["--ranks", "3", "-K", "5", "--host-events"], ["--ranks", "2", "--device", "0,0", "-t", "3"],
                                   # gathered output beyond --rank-buffer waits in a temporary file: none / a few hundred bytes in memory
                                   ["--ranks", "3", "--rank-buffer", "0"], ["--ranks", "2", "-K", "6", "--rank-buffer", "300"]])
def test_cli_ranks_on_random_signal_goldens(name, extra, models):
    """40-read compressed files, ragged batches inside every rank (-K 3/5/7 against 10-20 reads per rank)."""
    k, fasta, blow5, *args = open(os.path.join(GOLD, "random", name + ".args")).read().split("\n")
    cmd = [BIN, "dtw", "--kmer-model", models[int(k)], "--verbose", "0", *args, *extra, os.path.join(GOLD, "data", fasta),
           os.path.join(GOLD, "random", blow5)]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == open(os.path.join(GOLD, "random", name + ".out")).read()


@pytest.mark.parametrize("name", ["rnd_dna_sam", "rnd_rna", "rnd_dna"])
def test_cli_shards_and_read_ranges_concatenate(name, models):
    """What a launcher of its own would do (one process per GPU, any scheduler): `--shard r/G` per rank, `--no-header` on every
    rank but the first, outputs concatenated in rank order; and the same with explicit record ranges."""
    k, fasta, blow5, *args = open(os.path.join(GOLD, "random", name + ".args")).read().split("\n")
    want = open(os.path.join(GOLD, "random", name + ".out")).read()
    base = [BIN, "dtw", "--kmer-model", models[int(k)], "--verbose", "0", *args]
    files = [os.path.join(GOLD, "data", fasta), os.path.join(GOLD, "random", blow5)]

    def run(*extra):
        r = subprocess.run([*base, *extra, *files], capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        return r.stdout.decode()
    G = 3
    assert "".join(run("--shard", f"{r}/{G}", *(["--no-header"] if r else [])) for r in range(G)) == want
    cuts = [0, 1, 17, 40]
    parts = [run("--read-range", f"{a}:{b}", *(["--no-header"] if i else [])) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
    assert "".join(parts) == want
    assert run("--read-range", "17:", "--no-header") == parts[2]
    assert run("--read-range", "40:", "--no-header") == ""


def test_cli_ranks_report_and_failures(models, tmp_path):
    """--verbose 3: one line per rank (wall, bytes gathered) on stderr; a rank that fails makes the run fail; options that
    contradict the sharding are refused before anything starts."""
    import re
    c = load_case("dna_default")
    r = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "3", "--ranks", "2", "-t", "4", c["fasta"], c["blow5"]], capture_output=True, timeout=300)
    assert r.returncode == 0 and r.stdout.decode() == c["out_text"], r.stderr.decode()
    lines = re.findall(r"\[dtw_main\] rank (\d)/2 \(device 0, 2 host threads\): done after ([0-9.]+) sec, (\d+) bytes of output gathered", r.stderr.decode())
    assert [l[0] for l in lines] == ["0", "1"], r.stderr.decode()
    spilled = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "3", "--ranks", "2", "--rank-buffer", "100", c["fasta"], c["blow5"]],
                             capture_output=True, timeout=300)
    assert spilled.returncode == 0 and spilled.stdout.decode() == c["out_text"], spilled.stderr.decode()
    assert [l[2] for l in re.findall(r"rank (\d)/2 \(([^)]*)\): done after [0-9.]+ sec, (\d+) bytes", spilled.stderr.decode())] == [l[2] for l in lines]
    # -o FILE: the ranks and the supervisor share the one open file (rank 0 writes first, the gathered output follows)
    out = str(tmp_path / "ranks.paf")
    to_file = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "0", "--ranks", "3", "-o", out, c["fasta"], c["blow5"]], capture_output=True, timeout=300)
    assert to_file.returncode == 0 and to_file.stdout == b"" and open(out).read() == c["out_text"], to_file.stderr.decode()
    few = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "1", "--ranks", "3", "-t", "3", c["fasta"], c["blow5"]], capture_output=True, timeout=300)
    assert few.returncode == 0 and few.stdout.decode() == c["out_text"] and "1 host thread(s) per rank" in few.stderr.decode()
    for extra, msg in ((["--ranks", "2", "--shard", "0/2"], "cannot be combined"), (["--ranks", "2", "--read-range", "0:3"], "cannot be combined"),
                       (["--shard", "2/2"], "0 <= r < G"), (["--read-range", "5"], "A:B"), (["--ranks", "0"], "1..64"),
                       (["--ranks", "2", "--debug-break", "1"], "--ranks 1")):
        bad = subprocess.run([BIN, "dtw", "--kmer-model", models[6], *extra, c["fasta"], c["blow5"]], capture_output=True, timeout=60)
        assert bad.returncode != 0 and msg in bad.stderr.decode(), (extra, bad.stderr.decode())
    # the second rank of two meets a file cut inside a record: the run fails as a whole
    data = open(c["blow5"], "rb").read()
    cut = str(tmp_path / "cut.blow5")
    open(cut, "wb").write(data[:len(data) * 3 // 4])
    bad = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--ranks", "2", c["fasta"], cut], capture_output=True, timeout=120)
    assert bad.returncode != 0 and "rank of the sharded run failed" in bad.stderr.decode()


def test_cli_errors_like_reference(models):
    c = load_case("dna_default")
    for extra, msg in ((["--dtw-std"], "only available for RNA"), (["-p", "-1"], "auto query start")):
        r = subprocess.run([BIN, "dtw", "--kmer-model", models[6], *extra, c["fasta"], c["blow5"]], capture_output=True)
        assert r.returncode != 0 and msg in r.stderr.decode()
    r = subprocess.run([BIN, "dtw", c["fasta"], c["blow5"]], capture_output=True)
    assert r.returncode != 0 and "--kmer-model" in r.stderr.decode()


@pytest.mark.parametrize("extra", [[], ["--host-events"]])
def test_cli_profile_cpu_prints_the_reference_stage_timers(models, extra):
    """--profile-cpu=yes (src/dtw_main.c:213-214): stage by stage, with the reference's Parse / Events / Normalise / DTW
    lines on stderr (src/dtw_main.c:336-341); the output itself does not change."""
    import re
    c = load_case("dna_default")
    cmd = [BIN, "dtw", "--kmer-model", models[6], "--profile-cpu=yes", *extra, c["fasta"], c["blow5"]]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]
    err = r.stderr.decode()
    t = {}
    for key in ("Parse", "Events", "Normalise", "DTW"):
        m = re.search(r"\[dtw_main\]     - %s time: ([0-9.]+) sec" % key, err)
        assert m, (key, err)
        t[key] = float(m.group(1))
    assert re.search(r"\[dtw_main\] Data processing time: [0-9.]+ sec\n", err)
    assert t["DTW"] > 0 and t["Events"] >= 0  # (five reads: the host stages round to 0.000)
    # without the switch the four lines are not printed
    r = subprocess.run([BIN, "dtw", "--kmer-model", models[6], *extra, c["fasta"], c["blow5"]], capture_output=True, timeout=300)
    assert r.returncode == 0 and "- Parse time" not in r.stderr.decode() and r.stdout.decode() == c["out_text"]


def test_cli_verbose_4_reports_where_the_main_thread_waited(models):
    """--verbose 4: the timeline of a run on stderr -- initialisation, per batch lines, and (round 3) how long the main thread
    waited for a free device context and for the printer, which is how a run tells whether the host threads or the device set its
    pace; stdout is the same PAF."""
    import re
    c = load_case("dna_default")
    r = subprocess.run([BIN, "dtw", "--kmer-model", models[6], "--verbose", "4", "-K", "2", c["fasta"], c["blow5"]], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == c["out_text"]
    err = r.stderr.decode()
    assert re.search(r"\[dtw_main::[0-9.]+\] initialised: input [0-9.]+ s, model \+ reference events [0-9.]+ s, 2 device context\(s\) [0-9.]+ s", err), err
    m = re.search(r"main thread waited ([0-9.]+) sec for a free device context and ([0-9.]+) sec for the printer; page-locked staging \(re\)allocated in ([0-9.]+) sec", err)
    assert m, err
    assert all(float(x) >= 0 for x in m.groups())
    assert err.count("Entries") >= 6  # three batches of 2 + 2 + 1 reads: a `loaded` and a `processed` line each
