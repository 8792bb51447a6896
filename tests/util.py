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


def _svb_zd(raw):
    """int16 samples -> u32 n | StreamVByte keys | little-endian bytes of the zig-zag deltas (as tools/make_blow5.py)."""
    import struct
    x = raw.astype(np.int32)
    d = np.diff(x, prepend=np.int32(0))
    z = ((d << 1) ^ (d >> 31)).astype(np.uint32)
    nb = np.where(z < (1 << 8), 1, np.where(z < (1 << 16), 2, np.where(z < (1 << 24), 3, 4))).astype(np.uint8)
    n = len(z)
    codes = np.zeros((n + 3) // 4 * 4, np.uint8)
    codes[:n] = nb - 1
    keys = (codes[0::4] | (codes[1::4] << 2) | (codes[2::4] << 4) | (codes[3::4] << 6)).astype(np.uint8)
    b = z.view(np.uint8).reshape(n, 4)
    return struct.pack("<I", n) + keys.tobytes() + b[np.arange(4)[None, :] < nb[:, None]].tobytes()


def write_blow5(path, reads, attrs=(("experiment_type", "genomic_dna"), ("sequencing_kit", "unknown")), compress=False):
    """Minimal BLOW5 writer for synthetic test inputs: reads = [(read_id, digitisation, offset, range, sampling_rate,
    int16 samples)]; compress: zlib records + svb-zd signals.  Layout as tools/make_blow5.py / host/blow5.hpp."""
    import struct
    import zlib
    text = "".join(f"@{k}\t{v}\n" for k, v in attrs)
    text += "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
    text += "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    press = 1 if compress else 0
    hdr = b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([press]) + struct.pack("<I", 1) + bytes([press])
    hdr += b"\0" * (64 - len(hdr)) + struct.pack("<I", len(text)) + text.encode()
    with open(path, "wb") as out:
        out.write(hdr)
        for rid, dig, off, rng_, rate, raw in reads:
            name = rid.encode()
            raw = np.ascontiguousarray(raw, np.int16)
            body = _svb_zd(raw) if compress else raw.tobytes()
            payload = struct.pack("<H", len(name)) + name + struct.pack("<I4dQ", 0, dig, off, rng_, rate, len(body) if compress else len(raw)) + body
            if compress:
                payload = zlib.compress(payload, 6)
            out.write(struct.pack("<Q", len(payload)) + payload)
        out.write(b"5WOLB")


def visible_gpus():
    """GPUs this process can see, without initialising any of them (torch.cuda.device_count() does not, on this image)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def device_lists():
    """Device lists for the sharding tests.  One physical GPU listed two and three times always (several shards on one GPU,
    each with its own stream, scratch and host thread); wherever the box has MORE than one GPU the lists widen by themselves:
    every device once, every device in reverse order, and the first two -- so the first multi-GPU box that runs the suite
    exercises hipMemcpyPeer between distinct devices and one host thread per device without a code change."""
    lists = [[0, 0], [0, 0, 0]]
    g = visible_gpus()
    if g > 1:
        lists += [list(range(g)), list(range(g - 1, -1, -1)), [0, 1]]
    return lists


def distinct_device_list():
    """[0 .. G-1] when the box has G > 1 GPUs, else None (tests that only make sense across devices skip)."""
    g = visible_gpus()
    return list(range(g)) if g > 1 else None
