"""Shared helpers for the tests (not product code)."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "cases", "*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLD, "cases", name + ".npz"))
    c = {k: z[k] for k in z.files}
    c["name"] = name
    c["flag"] = int(c["flag"])
    c["k"] = int(c["k"])
    c["fasta"] = os.path.join(GOLD, "data", str(c["fasta"]))
    c["blow5"] = os.path.join(GOLD, "data", str(c["blow5"]))
    c["levels"] = np.fromfile(os.path.join(GOLD, "models", f"syn{c['k']}.f32"), np.float32)
    c["out_text"] = open(os.path.join(GOLD, "cases", name + ".out")).read()
    args = [str(a) for a in c["args"]]
    c["query_size"] = int(args[args.index("-q") + 1]) if "-q" in args else 250
    c["prefix_size"] = int(args[args.index("-p") + 1]) if "-p" in args else 50
    c["sam"] = "--sam" in args
    return c


def paf_lines_from_results(O, c, res, ref_names, ref_seq_lengths):
    """Format PAF with the oracle's writer for the valid reads of golden case c given result rows."""
    lines = []
    vi = 0
    for i, rid in enumerate(c["read_ids"]):
        if not c["read_valid"][i]:
            continue
        r = res[vi]
        qs, qe = int(c["qstart"][i]), int(c["qend"][i])
        start_raw = int(c["ev_start_first"][vi])
        end_raw = int(c["ev_start_last"][vi]) + int(c["ev_len_last"][vi])
        lines.append(O.paf_row(r, str(rid), str(ref_names[int(r["rid"])]), start_raw, end_raw, (qe - 1) - qs,
                               int(c["len_raw"][i]), int(ref_seq_lengths[int(r["rid"])])))
        vi += 1
    return "".join(lines)
