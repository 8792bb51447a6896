#!/usr/bin/env python3
"""Device-side record decoding (inflate + fields + StreamVByte) per batch size: decode_ms of sfa_align_blow5 next to the
event-detection and alignment stages of the same call.  Run on the GPU box:  python tools/blow5_decode_bench.py [K ...]"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sigfish_amd as S  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    ks = [int(a) for a in sys.argv[1:]] or [512, 4096, 16384]
    d = tempfile.mkdtemp()
    path = os.path.join(d, "c.blow5")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_blow5.py"), os.path.join(GOLD, "data", "sp1_dna.blow5"), path,
                    "--copies", str(max(ks) // 5 + 1), "--compress", "--jobs", "8"], check=True, capture_output=True)
    b = open(path, "rb").read()
    (hl,) = struct.unpack_from("<I", b, 64)
    p = 68 + hl
    recs = []
    while b[p:p + 5] != b"5WOLB":
        (sz,) = struct.unpack_from("<Q", b, p)
        recs.append(b[p + 8:p + 8 + sz])
        p += 8 + sz
    lv = np.fromfile(os.path.join(GOLD, "models", "syn6.f32"), np.float32)
    ref = S.RefModel.from_fasta(os.path.join(GOLD, "data", "nCoV-2019.reference.fasta"), lv, 6, 0, 250)
    with S.Aligner(ref, 0, device=0) as al:
        for k in ks:
            rs = recs[:k]
            off = np.concatenate([[0], np.cumsum([len(r) for r in rs])]).astype(np.int64)
            blob = b"".join(rs)
            for _ in range(3):
                al.align_blow5(blob, off, True, True)
            pr = al.profile()
            print(f"K {k}: decode {pr['decode_ms']:.3f} ms, events {pr['events_ms']:.3f} ms, normalise {pr['normalise_ms']:.3f} ms, "
                  f"alignment {pr['total_ms']:.3f} ms; {len(blob) / 1e6:.1f} MB of records, fallbacks {pr['blow5_fallbacks']}")


if __name__ == "__main__":
    main()
