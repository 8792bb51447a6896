"""The k-mer model reader (host/refio.cpp: read_kmer_model) accepts and refuses what the reference's read_model does.

`src/model.c` cannot be compiled here (it includes the missing blob `src/model.h`), so this is a table of cases written from
reading `src/model.c:38-131`, each row citing the lines that decide it:

  :60-66   a line is skipped iff it starts with '#', '\\n' or '\\r', or IS one of three header spellings (newline included);
  :69-84   among the skipped lines, "#k<ws>INT" sets k (<= 0 and > MAX_KMER_SIZE = 9 are fatal); it may come again;
  :86-100  every other line is a table row: sscanf("%s\\t%f\\t%f"), the row is COUNTED whatever sscanf returned, a return
           other than 3 is logged and the run goes on;
  :101-106 more rows than 4^k: fatal;   :111-116 fewer rows than 4^k at the end of the file: fatal;
  :40-41   no "#k" line: k = MAX_KMER_SIZE = 9.
"""
import itertools

import numpy as np
import pytest

import sigfish_amd as S

H5 = "kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n"
H6 = "kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\tweight\n"
H7 = "kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\tig_lambda\tweight\n"


def rows(k, n=None, fmt="%s\t%.4f\t1.5\t1.0\t1.0\n", base=50.0):
    out = []
    for i, kmer in enumerate(itertools.product("ACGT", repeat=k)):
        if n is not None and i >= n:
            break
        out.append(fmt % ("".join(kmer), base + i))
    return out


def levels(k, base=50.0):
    return (base + np.arange(4 ** k)).astype(np.float32)


# (name, file text, outcome) -- outcome: ("ok", k, expected levels | None, n_warnings) or ("fail", message part)
CASES = [
    ("plain_k2", "#k\t2\n" + H5 + "".join(rows(2)), ("ok", 2, levels(2), 0)),
    ("header_with_weight", "#k\t2\n" + H6 + "".join(rows(2)), ("ok", 2, levels(2), 0)),
    ("header_with_ig_lambda", "#k\t2\n" + H7 + "".join(rows(2)), ("ok", 2, levels(2), 0)),
    ("no_header_line", "#k\t2\n" + "".join(rows(2)), ("ok", 2, levels(2), 0)),
    ("comments_and_blank_lines_anywhere", "#ont_model\tx\n\n#k\t2\n" + H5 + "".join(rows(2)[:5]) + "# comment\n\r\n" + H5 + "".join(rows(2)[5:]) + "\n",
     ("ok", 2, levels(2), 0)),
    ("k_given_with_a_space", "#k 2\n" + "".join(rows(2)), ("ok", 2, levels(2), 0)),          # scanf's \t matches any white space
    ("k_glued_is_not_a_k_line", "#k2\n" + "".join(rows(2)), ("fail", "prematurely ended")),   # a comment; k stays 9
    ("k_twice_last_one_counts", "#k\t3\n#k\t2\n" + "".join(rows(2)), ("ok", 2, levels(2), 0)),
    ("k_line_after_the_rows", "".join(rows(2)) + "#k\t2\n", ("ok", 2, levels(2), 0)),           # the count is checked at the end
    ("two_fields_more_than_needed_is_fine", "#k\t1\n" + "".join(rows(1, fmt="%s\t%.4f\t1.5\t1\t1\t7\t8\n")), ("ok", 1, levels(1), 0)),
    ("three_fields_only", "#k\t1\n" + "".join(rows(1, fmt="%s\t%.4f\t1.5\n")), ("ok", 1, levels(1), 0)),
    ("fields_separated_by_spaces", "#k\t1\n" + "".join(rows(1, fmt="%s %.4f 1.5\n")), ("ok", 1, levels(1), 0)),
    ("last_line_without_newline", "#k\t1\n" + "".join(rows(1))[:-1], ("ok", 1, levels(1), 0)),
    ("kmer_names_are_not_checked", "#k\t1\nXX\t50\t1\nA\t51\t1\nA\t52\t1\nZ\t53\t1\n", ("ok", 1, levels(1), 0)),    # order in the file is the rank
    # ---- rows that do not parse are logged, counted, and the run goes on (:98-100) ----
    ("row_with_two_fields_keeps_its_mean", "#k\t1\nA\t50\t1\nC\t51\nG\t52\t1\nT\t53\t1\n", ("ok", 1, levels(1), 1)),
    ("row_with_one_field_counts", "#k\t1\nA\t50\t1\nC\nG\t52\t1\nT\t53\t1\n", ("ok", 1, np.float32([50, 0, 52, 53]), 1)),
    ("text_where_a_number_should_be", "#k\t1\nA\t50\t1\nC\tabc\t1\nG\t52\t1\nT\t53\t1\n", ("ok", 1, np.float32([50, 0, 52, 53]), 1)),
    # ---- header spellings are exact (:63-65): anything else that starts with "kmer" is a table row ----
    ("unknown_header_is_a_row_too_many", "#k\t1\nkmer\tlevel_mean\tlevel_stdv\n" + "".join(rows(1)), ("fail", "too many entries")),
    ("crlf_header_is_a_row_too_many", "#k\t1\n" + H5[:-1] + "\r\n" + "".join(rows(1)), ("fail", "too many entries")),
    ("header_without_newline_at_eof_is_a_row", "#k\t1\n" + "".join(rows(1)) + H5[:-1], ("fail", "too many entries")),
    ("unknown_header_takes_a_rows_place", "#k\t1\nkmer\tmean\n" + "".join(rows(1)[:3]), ("ok", 1, np.float32([0, 50, 51, 52]), 1)),
    ("crlf_rows_are_fine", "#k\t1\n" + "".join(r[:-1] + "\r\n" for r in rows(1)), ("ok", 1, levels(1), 0)),
    # ---- counts (:101-116) ----
    ("one_row_short", "#k\t2\n" + "".join(rows(2, 15)), ("fail", "prematurely ended. Expected 16 kmers in the model, but file had only 15")),
    ("one_row_over", "#k\t2\n" + "".join(rows(2)) + "AA\t1\t1\n", ("fail", "too many entries. Expected 16 kmers")),
    ("empty_file", "", ("fail", "prematurely ended. Expected 262144 kmers in the model, but file had only 0")),
    ("no_k_line_means_k9", "".join(rows(2)), ("fail", "Expected 262144 kmers")),
    # ---- #k range (:72-79) ----
    ("k_zero", "#k\t0\n", ("fail", "(#k\t0) in file")),
    ("k_negative", "#k\t-3\n", ("fail", "is invalid")),
    ("k_ten", "#k\t10\n", ("fail", "larger than MAX_KMER_SIZE (9)")),
    ("k_not_a_number_is_a_comment", "#k\tsix\n#k\t1\n" + "".join(rows(1)), ("ok", 1, levels(1), 0)),
]


@pytest.mark.parametrize("name,text,outcome", CASES, ids=[c[0] for c in CASES])
def test_acceptance_table(name, text, outcome, tmp_path):
    p = tmp_path / (name + ".model")
    p.write_bytes(text.encode())
    if outcome[0] == "fail":
        with pytest.raises(S.SfaError) as e:
            S.read_kmer_model(p)
        assert outcome[1] in str(e.value), str(e.value)
        return
    _, k, want, n_warn = outcome
    warns = []
    got, kk = S.read_kmer_model(p, warnings=warns)
    assert kk == k
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
    assert len(warns) == n_warn, warns
    assert all("is corrupted at line" in w for w in warns)


def test_k9_table_without_a_k_line(tmp_path):
    """The default: 262 144 rows and no "#k" line is a 9-mer model (what the reference's own r10 files look like)."""
    p = tmp_path / "k9.model"
    with open(p, "w") as f:
        f.write(H5)
        f.writelines("%s\t%.3f\t2.0\n" % ("".join(kmer), 60 + (i % 977) * 0.061) for i, kmer in enumerate(itertools.product("ACGT", repeat=9)))
    got, k = S.read_kmer_model(p)
    want = np.float32([float("%.3f" % (60 + (i % 977) * 0.061)) for i in range(4 ** 9)])
    assert k == 9 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_missing_file(tmp_path):
    with pytest.raises(S.SfaError, match="cannot open"):
        S.read_kmer_model(tmp_path / "nope.model")
