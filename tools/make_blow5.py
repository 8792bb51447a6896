#!/usr/bin/env python3
"""make_blow5.py -- write a large BLOW5 file by replicating the reads of a fixture (SURVEY.md §8d "end-to-end synthetic
inputs"), to measure the `sigfish-amd dtw` command line end to end.  Uncompressed by default; --compress writes what
real files use: StreamVByte zig-zag-delta signals inside zlib records.

    python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 /tmp/big.blow5 --copies 4000

Layout written (slow5 spec 0.2.0, see sigfish_amd/csrc/host/blow5.hpp): magic, version 0.2.0, record_press none,
one read group, signal_press none, ascii header with the source's attributes and the primary columns only, records
[u64 size][u16 id_len, id, u32 group, 4 x f64, u64 n, int16 samples], EOF marker.
"""
import argparse
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigfish_amd as S  # noqa: E402


def svb_zd(raw):
    """int16 samples -> u32 n | StreamVByte keys (2 bits per value: bytes-1) | little-endian data bytes of the zig-zag deltas."""
    x = raw.astype(np.int32)
    d = np.diff(x, prepend=np.int32(0))
    z = ((d << 1) ^ (d >> 31)).astype(np.uint32)
    nb = np.where(z < (1 << 8), 1, np.where(z < (1 << 16), 2, np.where(z < (1 << 24), 3, 4))).astype(np.uint8)
    n = len(z)
    codes = np.zeros((n + 3) // 4 * 4, np.uint8)
    codes[:n] = nb - 1
    keys = (codes[0::4] | (codes[1::4] << 2) | (codes[2::4] << 4) | (codes[3::4] << 6)).astype(np.uint8)
    b = z.view(np.uint8).reshape(n, 4)  # little-endian bytes of every value
    mask = np.arange(4)[None, :] < nb[:, None]
    return struct.pack("<I", n) + keys.tobytes() + b[mask].tobytes()


def _records(args):
    """records of copies [c0, c1) as one bytes object (worker of --jobs)"""
    reads, sig, c0, c1, rec_zlib, sig_svb = args
    out = []
    for c in range(c0, c1):
        for (rid, meta, raw), body in zip(reads, sig):
            name = f"{rid}_{c}".encode()
            payload = struct.pack("<H", len(name)) + name + struct.pack("<I4dQ", 0, meta["digitisation"], meta["offset"], meta["range"],
                                                                      meta["sampling_rate"], len(body) if sig_svb else len(raw)) + body
            if rec_zlib:
                payload = zlib.compress(payload, 6)
            out.append(struct.pack("<Q", len(payload)) + payload)
    return b"".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--copies", type=int, default=1000)
    ap.add_argument("--compress", action="store_true", help="zlib records + svb-zd signals (as real BLOW5 files)")
    ap.add_argument("--record-press", choices=["none", "zlib"], default=None, help="record compression alone")
    ap.add_argument("--signal-press", choices=["none", "svb-zd"], default=None, help="signal compression alone")
    ap.add_argument("--jobs", type=int, default=1, help="worker processes (zlib of every record is what takes the time)")
    ap.add_argument("--ascii", action="store_true", help="write SLOW5 ASCII (the text twin of the format; doubles with repr(), so they read back exactly)")
    ap.add_argument("--keep-ids", action="store_true", help="with --ascii --copies 1: the reads keep their ids (no _0 suffix)")
    ap.add_argument("--aux", type=int, default=0, help="with --ascii: that many auxiliary columns (uint64_t, value = the copy index)")
    a = ap.parse_args()
    f = S.Blow5File(a.src)
    reads = list(f)
    attrs = [("experiment_type", f.attr("experiment_type") or "genomic_dna"), ("sequencing_kit", f.attr("sequencing_kit") or "unknown")]
    text = "".join(f"@{k}\t{v}\n" for k, v in attrs)
    text += "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
    text += "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    if a.ascii:
        with open(a.dst, "w") as out:
            out.write("#slow5_version\t0.2.0\n#num_read_groups\t1\n" + "".join(f"@{k}\t{v}\n" for k, v in attrs))
            out.write("#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*" + "\tuint64_t" * a.aux + "\n")
            out.write("#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal"
                      + "".join(f"\taux{i}" for i in range(a.aux)) + "\n")

            def num(x):  # slow5lib takes digits, '.' and '-' only: no exponent, and "8192" rather than "8192.0" as slow5tools prints it
                t = repr(float(x))
                assert "e" not in t and "n" not in t, t
                return t[:-2] if t.endswith(".0") else t
            sigs = [",".join(map(str, raw.tolist())) for _, _, raw in reads]
            for c in range(a.copies):
                for (rid, meta, raw), sig in zip(reads, sigs):
                    out.write((rid if a.keep_ids and a.copies == 1 else f"{rid}_{c}") + f"\t0\t{num(meta['digitisation'])}\t{num(meta['offset'])}\t{num(meta['range'])}\t"
                              f"{num(meta['sampling_rate'])}\t{len(raw)}\t{sig}" + f"\t{c}" * a.aux + "\n")
        print(f"{a.dst}: {a.copies * len(reads)} reads, {os.path.getsize(a.dst) / 1e6:.1f} MB (SLOW5 ASCII)")
        return
    rec_zlib = (a.record_press == "zlib") if a.record_press else a.compress
    sig_svb = (a.signal_press == "svb-zd") if a.signal_press else a.compress
    hdr = b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([1 if rec_zlib else 0]) + struct.pack("<I", 1) + bytes([1 if sig_svb else 0])
    hdr += b"\0" * (64 - len(hdr)) + struct.pack("<I", len(text)) + text.encode()
    n = 0
    with open(a.dst, "wb") as out:
        out.write(hdr)
        sig = [svb_zd(raw) if sig_svb else raw.tobytes() for _, _, raw in reads]
        step = max(1, min(2000, a.copies // max(4 * a.jobs, 1) or 1))
        tasks = [(reads, sig, c0, min(c0 + step, a.copies), rec_zlib, sig_svb) for c0 in range(0, a.copies, step)]
        if a.jobs > 1 and len(tasks) > 1:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(a.jobs) as pool:
                for blob in pool.imap(_records, tasks):  # in order: read ids stay in file order
                    out.write(blob)
        else:
            for t in tasks:
                out.write(_records(t))
        n = a.copies * len(reads)
        out.write(b"5WOLB")
    print(f"{a.dst}: {n} reads, {os.path.getsize(a.dst) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
