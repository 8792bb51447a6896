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


def write_blow5(path, reads, attrs=(("experiment_type", "genomic_dna"), ("sequencing_kit", "unknown"))):
    """Minimal uncompressed BLOW5 writer for synthetic test inputs: reads = [(read_id, digitisation, offset, range,
    sampling_rate, int16 samples)].  Layout as tools/make_blow5.py / sigfish_amd/csrc/host/blow5.hpp."""
    import struct
    text = "".join(f"@{k}\t{v}\n" for k, v in attrs)
    text += "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
    text += "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    hdr = b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([0]) + struct.pack("<I", 1) + bytes([0])
    hdr += b"\0" * (64 - len(hdr)) + struct.pack("<I", len(text)) + text.encode()
    with open(path, "wb") as out:
        out.write(hdr)
        for rid, dig, off, rng_, rate, raw in reads:
            name = rid.encode()
            raw = np.ascontiguousarray(raw, np.int16)
            payload = struct.pack("<H", len(name)) + name + struct.pack("<I4dQ", 0, dig, off, rng_, rate, len(raw)) + raw.tobytes()
            out.write(struct.pack("<Q", len(payload)) + payload)
        out.write(b"5WOLB")
