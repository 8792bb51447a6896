"""SLOW5 ASCII, the text twin of BLOW5 (slow5_open takes either, slow5lib/src/slow5.c:4219-4229).  The reader
hands out the same records from a text file as from the binary file with the same reads (fields bit for bit, samples equal),
shards a text file by byte slices like a binary one, and refuses what slow5lib's parser refuses (slow5.c:2643-2790,
slow5_misc.c:103-156) -- checked against the compiled reference where oracle/_ref exists."""
import os
import subprocess
import sys

import numpy as np
import pytest

import sigfish_amd as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")  # the compiled reference's own pipeline (checker only)


def make(src, dst, *flags):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", src), str(dst), *flags],
                   check=True, capture_output=True)
    return str(dst)


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("slow5")
    return {"dna_txt": make("sp1_dna.blow5", d / "dna.slow5", "--copies", "3", "--ascii"),
            "dna_bin": make("sp1_dna.blow5", d / "dna.blow5", "--copies", "3", "--compress"),
            "rna_txt": make("sequin_rna.blow5", d / "rna.slow5", "--copies", "1", "--ascii", "--aux", "2"),
            "rna_bin": make("sequin_rna.blow5", d / "rna.blow5", "--copies", "1"), "dir": d}


def records(path, shard=None):
    f = S.Blow5File(path)
    if shard:
        f.select_shard(*shard)
    return list(f)


def same(a, b):
    return len(a) == len(b) and all(x[0] == y[0] and x[1] == y[1] and np.array_equal(x[2], y[2]) for x, y in zip(a, b))


@pytest.mark.parametrize("kind", ["dna", "rna"])
def test_text_and_binary_files_hold_the_same_records(files, kind):
    txt, bin_ = records(files[kind + "_txt"]), records(files[kind + "_bin"])
    assert len(txt) == (15 if kind == "dna" else 8) and same(txt, bin_)
    f = S.Blow5File(files[kind + "_txt"])
    assert f.attr("experiment_type") == ("genomic_dna" if kind == "dna" else "rna") and f.attr("no_such_attribute") is None


@pytest.mark.parametrize("G", [2, 3, 7, 40])
def test_shards_of_a_text_file_are_every_record_once_in_order(files, G):
    whole = records(files["dna_txt"])
    parts = [records(files["dna_txt"], (r, G)) for r in range(G)]
    assert same([x for p in parts for x in p], whole)
    assert same(list(S.Blow5File(files["dna_txt"]).select_records(4, 7)), whole[4:11])


def reference_accepts(path):
    r = subprocess.run([REF_DRIVER, "--model", os.path.join(GOLD, "models", "syn6.f32"), "--kmer", "6", "-t", "2",
                        os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), path], capture_output=True, timeout=300)
    return r.returncode == 0


def edit(files, name, fn):
    """the DNA text file with fn(header lines, record lines) applied; both lists hold lines without their newline"""
    lines = open(files["dna_txt"]).read().split("\n")[:-1]
    n_hdr = next(i for i, l in enumerate(lines) if l.startswith("#read_id")) + 1
    hdr, recs = lines[:n_hdr], [l.split("\t") for l in lines[n_hdr:]]
    tail = fn(hdr, recs)
    path = str(files["dir"] / (name + ".slow5"))
    open(path, "w").write("\n".join(hdr + ["\t".join(r) for r in recs]) + ("\n" if tail is None else tail))
    return path


def set_col(i, value, rec=1):
    def fn(hdr, recs):
        recs[rec][i] = value(recs[rec][i]) if callable(value) else value
    return fn


def set_hdr(i, value):
    def fn(hdr, recs):
        hdr[i] = value(hdr[i]) if callable(value) else value
    return fn


REFUSED = {
    "group_leading_zero": set_col(1, "00"),
    "group_signed": set_col(1, "+0"),
    "group_empty": set_col(1, ""),
    "group_too_large": set_col(1, "4294967296"),
    "digitisation_exponent": set_col(2, "8.192e3"),
    "offset_empty": set_col(3, ""),
    "range_text": set_col(4, "nan"),
    "length_leading_zero": set_col(6, lambda v: "0" + v),
    "one_sample_fewer": set_col(7, lambda v: v.rsplit(",", 1)[0]),
    "sample_out_of_range": set_col(7, lambda v: "32768," + v.split(",", 1)[1]),
    "sample_leading_zero": set_col(7, lambda v: "007," + v.split(",", 1)[1]),
    "sample_empty": set_col(7, lambda v: "," + v.split(",", 1)[1]),
    "sample_text": set_col(7, lambda v: "x," + v.split(",", 1)[1]),
    "no_comma": set_col(7, "5"),
    "seven_columns": lambda hdr, recs: recs[2].pop(),
    "aux_not_announced": lambda hdr, recs: recs[2].append("17"),
    "version_newer": set_hdr(0, "#slow5_version\t1.1.0"),
    "version_short": set_hdr(0, "#slow5_version\t0.2"),
    "groups_zero": set_hdr(1, "#num_read_groups\t0"),
    "no_types_line": lambda hdr, recs: hdr.pop(-2),
    "types_renamed": set_hdr(-2, lambda v: v.replace("uint64_t", "uint32_t")),
    "names_renamed": set_hdr(-1, lambda v: v.replace("range", "rng")),
    "aux_type_without_name": set_hdr(-2, lambda v: v + "\tuint8_t"),
    "stray_header_line": lambda hdr, recs: hdr.insert(2, "experiment_type\tgenomic_dna"),
}


@pytest.mark.parametrize("name", sorted(REFUSED))
def test_what_slow5lib_refuses_is_refused(files, name):
    path = edit(files, name, REFUSED[name])
    with pytest.raises(S.SfaError):
        records(path)
    if os.path.exists(REF_DRIVER):  # ... and the compiled reference refuses every one of them too
        assert not reference_accepts(path), "the compiled reference accepts this file"


def test_more_samples_than_announced_and_a_missing_last_newline_are_errors(files):
    """Two cases where slow5lib has no check: it writes past its buffer on the first (slow5.c:2755-2764) and drops the last
    character of the file on the second (slow5.c:3214); here both are malformed files."""
    with pytest.raises(S.SfaError, match="more samples"):
        records(edit(files, "too_many", set_col(7, lambda v: v + ",1,2,3")))
    with pytest.raises(S.SfaError, match="newline"):
        records(edit(files, "no_newline", lambda hdr, recs: ""))
    assert len(records(edit(files, "header_only", lambda hdr, recs: recs.clear()))) == 0


def test_oddities_slow5lib_accepts_read_the_same(files):
    """strtol's reading of tokens that pass slow5_int_check (digits and '-' anywhere): "12-3" is 12, "-" is 0; doubles may
    carry several dots or dashes as far as strtod reads them.  A zero length leaves the signal column unread (slow5.c:2722-2725;
    the reference's own pipeline then aborts in free(), so that record is not shown to it)."""
    def fn(hdr, recs):
        recs[0][7] = "12-3,-," + recs[0][7].split(",", 2)[2]
        recs[2][3] = "14.5.7"
    path = edit(files, "oddities", fn)
    got, want = records(path), records(files["dna_txt"])
    assert list(got[0][2][:2]) == [12, 0] and np.array_equal(got[0][2][2:], want[0][2][2:])
    assert got[2][1]["offset"] == 14.5 and same(got[3:], want[3:]) and same(got[1:2], want[1:2])
    if os.path.exists(REF_DRIVER):
        assert reference_accepts(path)

    def empty(hdr, recs):
        recs[1][6], recs[1][7] = "0", "whatever"
    got = records(edit(files, "zero_length", empty))
    assert len(got[1][2]) == 0 and same(got[:1] + got[2:], want[:1] + want[2:])


def test_a_binary_name_with_text_inside_is_read_by_content(files):
    """slow5lib goes by the extension; the reader here by the first bytes, so a renamed file still opens"""
    path = str(files["dir"] / "renamed.blow5")
    open(path, "w").write(open(files["dna_txt"]).read())
    assert same(records(path), records(files["dna_txt"]))
    junk = str(files["dir"] / "junk.slow5")
    open(junk, "w").write("#slow5_versio\t0.2.0\n" + "x" * 100)
    with pytest.raises(S.SfaError, match="not a BLOW5 file"):
        records(junk)
