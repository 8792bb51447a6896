#!/usr/bin/env python3
"""make_blow5.py -- write a large UNCOMPRESSED BLOW5 file by replicating the reads of a fixture (SURVEY.md §8d
"end-to-end synthetic inputs"), to measure the `sigfish-amd dtw` command line end to end.

    python tools/make_blow5.py tests/golden/data/sp1_dna.blow5 /tmp/big.blow5 --copies 4000

Layout written (slow5 spec 0.2.0, see sigfish_amd/csrc/host/blow5.hpp): magic, version 0.2.0, record_press none,
one read group, signal_press none, ascii header with the source's attributes and the primary columns only, records
[u64 size][u16 id_len, id, u32 group, 4 x f64, u64 n, int16 samples], EOF marker.
"""
import argparse
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigfish_amd as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--copies", type=int, default=1000)
    a = ap.parse_args()
    f = S.Blow5File(a.src)
    reads = list(f)
    attrs = [("experiment_type", f.attr("experiment_type") or "genomic_dna"), ("sequencing_kit", f.attr("sequencing_kit") or "unknown")]
    text = "".join(f"@{k}\t{v}\n" for k, v in attrs)
    text += "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
    text += "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    hdr = b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([0]) + struct.pack("<I", 1) + bytes([0])
    hdr += b"\0" * (64 - len(hdr)) + struct.pack("<I", len(text)) + text.encode()
    n = 0
    with open(a.dst, "wb") as out:
        out.write(hdr)
        for c in range(a.copies):
            for rid, meta, raw in reads:
                name = f"{rid}_{c}".encode()
                payload = struct.pack("<H", len(name)) + name + struct.pack("<I4dQ", 0, meta["digitisation"], meta["offset"], meta["range"],
                                                                          meta["sampling_rate"], len(raw)) + raw.tobytes()
                out.write(struct.pack("<Q", len(payload)) + payload)
                n += 1
        out.write(b"5WOLB")
    print(f"{a.dst}: {n} reads, {os.path.getsize(a.dst) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
