"""oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of the CPU restatement in oracle/sdtw_oracle.c (built into oracle/_build/liboracle.so)
and, when present, of the compiled reference sources in oracle/_ref/ (this container only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (sigfish_amd/) never does.
"""
import ctypes as C
import gzip
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsigfish_ref.so")
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")
REF_BENCH = os.path.join(HERE, "_ref", "ref_bench")

RNA, DTW, INV, REF, END = 0x001, 0x002, 0x004, 0x010, 0x020

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


class Result(C.Structure):
    _fields_ = [("rid", C.c_int32), ("pos_st", C.c_int32), ("pos_end", C.c_int32), ("score", C.c_float),
                ("score2", C.c_float), ("strand", C.c_int8), ("mapq", C.c_uint8), ("valid", C.c_uint8),
                ("pad", C.c_uint8)]


RESULT_DTYPE = np.dtype([("rid", "<i4"), ("pos_st", "<i4"), ("pos_end", "<i4"), ("score", "<f4"), ("score2", "<f4"),
                         ("strand", "i1"), ("mapq", "u1"), ("valid", "u1"), ("pad", "u1")])


class RefC(C.Structure):
    _fields_ = [("num_ref", C.c_int32), ("ref_lengths", _i32p), ("ref_st_offset", _i32p),
                ("forward", C.POINTER(_f32p)), ("reverse", C.POINTER(_f32p))]


def build(ref=True):
    """Compile the oracle (and the reference build when /root/reference exists). Building != using."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        L.orc_subsequence.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.orc_std_dtw.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.orc_std_dtw.restype = C.c_float
        L.orc_subsequence_path.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _i32p, _i32p]
        L.orc_path_start.argtypes = [_f32p, C.c_int, C.c_int, C.c_int]
        L.orc_normalise.argtypes = [_f32p, C.c_uint64]
        L.orc_kmer_rank.argtypes = [C.c_char_p, C.c_uint32]
        L.orc_kmer_rank.restype = C.c_uint32
        L.orc_gen_ref_record.argtypes = [C.c_char_p, C.c_int32, _f32p, C.c_uint32, C.c_uint32, C.c_int32, _f32p,
                                         _f32p, _i32p]
        L.orc_gen_ref_record.restype = C.c_int32
        L.orc_dtw_single.argtypes = [_f32p, C.c_int32, C.POINTER(RefC), C.c_uint32, C.POINTER(Result)]
        L.orc_align_batch.argtypes = [_f32p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(RefC), C.c_uint32,
                                      C.c_int32, C.c_void_p]
        L.orc_mapq.argtypes = [C.c_float, C.c_float]
        L.orc_mapq.restype = C.c_uint8
        L.orc_paf_row.argtypes = [C.c_char_p, C.c_int, C.POINTER(Result), C.c_char_p, C.c_char_p, C.c_uint64,
                                  C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_query_window.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- kernel level
def subsequence(x, y):
    x, y = _f32(x), _f32(y)
    cost = np.empty((len(x), len(y)), np.float32)
    lib().orc_subsequence(_fp(x), _fp(y), len(x), len(y), _fp(cost))
    return cost


def std_dtw(x, y):
    x, y = _f32(x), _f32(y)
    cost = np.empty((len(x), len(y)), np.float32)
    lib().orc_std_dtw(_fp(x), _fp(y), len(x), len(y), _fp(cost))
    return cost


def subsequence_path(cost, starty):
    cost = _f32(cost)
    n, m = cost.shape
    px = np.empty(n + m + 2, np.int32)
    py = np.empty(n + m + 2, np.int32)
    k = lib().orc_subsequence_path(_fp(cost), n, m, starty, px.ctypes.data_as(_i32p), py.ctypes.data_as(_i32p))
    return px[:k].copy(), py[:k].copy()


def path_start(cost, starty):
    cost = _f32(cost)
    return lib().orc_path_start(_fp(cost), cost.shape[0], cost.shape[1], starty)


def normalise(v):
    v = _f32(v).copy()
    lib().orc_normalise(_fp(v), len(v))
    return v


def mapq(score, score2):
    return int(lib().orc_mapq(float(np.float32(score)), float(np.float32(score2))))


# ---------------------------------------------------------------- reference event arrays
def read_fasta(path):
    """[(name, sequence)] with kseq.h semantics: name = first word after '>', sequence lines concatenated."""
    op = gzip.open if path.endswith(".gz") else open
    recs, name, chunks = [], None, []
    with op(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(chunks)))
                name, chunks = line[1:].split()[0] if line[1:].split() else "", []
            elif name is not None:
                chunks.append("".join(line.split()))
    if name is not None:
        recs.append((name, "".join(chunks)))
    return recs


class RefSynth:
    """refsynth_t (src/sigfish.h:90-99) as numpy arrays."""

    def __init__(self, names, seq_lengths, ref_lengths, st_offset, forward, reverse):
        self.names, self.seq_lengths = names, np.asarray(seq_lengths, np.int32)
        self.ref_lengths = np.asarray(ref_lengths, np.int32)
        self.st_offset = np.asarray(st_offset, np.int32)
        self.forward, self.reverse = forward, reverse
        self.num_ref = len(names)
        self._keep = None

    def as_c(self):
        n = self.num_ref
        fa = (_f32p * n)(*[_fp(a) for a in self.forward])
        ra = (_f32p * n)(*[_fp(a) for a in self.reverse]) if self.reverse is not None else None
        r = RefC(n, self.ref_lengths.ctypes.data_as(_i32p), self.st_offset.ctypes.data_as(_i32p), fa,
                 C.cast(ra, C.POINTER(_f32p)) if ra is not None else None)
        self._keep = (fa, ra)
        return r


def gen_ref(records, levels, k, flag, query_size):
    """genref.c:86-241 over [(name, seq)]."""
    levels = _f32(levels)
    names, sl, rl, so, fw, rv = [], [], [], [], [], []
    rna = bool(flag & RNA)
    for name, seq in records:
        b = seq.encode()
        cap = max(len(b) + 1 - k, 1)
        f = np.zeros(cap, np.float32)
        r = np.zeros(cap, np.float32)
        off = C.c_int32(0)
        n = lib().orc_gen_ref_record(b, len(b), _fp(levels), k, flag, query_size, _fp(f), _fp(r), C.byref(off))
        names.append(name)
        sl.append(len(b))
        rl.append(n)
        so.append(off.value)
        fw.append(f[:n].copy())
        rv.append(r[:n].copy())
    return RefSynth(names, sl, rl, so, fw, None if rna else rv)


# ---------------------------------------------------------------- per-read / batch alignment
def dtw_single(events, ref, flag):
    ev = _f32(events)
    out = Result()
    rc = ref.as_c()
    lib().orc_dtw_single(_fp(ev), len(ev), C.byref(rc), flag, C.byref(out))
    return out


def align_batch(events, q_off, ref, flag, threads=1):
    """events: concatenated float32 event means (event order), q_off: int64[n+1]. -> structured array."""
    ev = _f32(events)
    q_off = np.ascontiguousarray(q_off, np.int64)
    n = len(q_off) - 1
    out = np.zeros(n, RESULT_DTYPE)
    rc = ref.as_c()
    lib().orc_align_batch(_fp(ev), q_off.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(rc), flag, threads,
                          out.ctypes.data_as(C.c_void_p))
    return out


def paf_row(res, read_id, rname, start_raw, end_raw, query_size, len_raw, rlength):
    r = Result(int(res["rid"]), int(res["pos_st"]), int(res["pos_end"]), float(res["score"]), float(res["score2"]),
               int(res["strand"]), int(res["mapq"]), int(res["valid"]), 0)
    buf = C.create_string_buffer(4096)
    n = lib().orc_paf_row(buf, 4096, C.byref(r), read_id.encode(), rname.encode(), start_raw, end_raw, query_size,
                          len_raw, rlength)
    return buf.raw[:n].decode()


def query_window(n_events, prefix, qsize, flag):
    a, b = C.c_int64(), C.c_int64()
    keep = lib().orc_query_window(n_events, prefix, qsize, flag, C.byref(a), C.byref(b))
    return keep, a.value, b.value


# ---------------------------------------------------------------- compiled reference (this container only)
_ref = None


def reference_lib():
    """libsigfish_ref.so (RTLD_LAZY: read_model/set_model are unresolved because model.c is unbuildable)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            return None
        L = C.CDLL(REF_SO, mode=os.RTLD_LAZY)
        L.subsequence.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.std_dtw.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p, C.c_int]
        L.std_dtw.restype = C.c_float

        class Path(C.Structure):
            _fields_ = [("k", C.c_int), ("px", _i32p), ("py", _i32p)]

        L.Path = Path
        L.subsequence_path.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.POINTER(Path)]
        L._libc = C.CDLL(None)
        L._libc.free.argtypes = [C.c_void_p]
        _ref = L
    return _ref


def ref_subsequence(x, y):
    x, y = _f32(x), _f32(y)
    cost = np.empty((len(x), len(y)), np.float32)
    reference_lib().subsequence(_fp(x), _fp(y), len(x), len(y), _fp(cost))
    return cost


def ref_std_dtw(x, y):
    x, y = _f32(x), _f32(y)
    cost = np.empty((len(x), len(y)), np.float32)
    reference_lib().std_dtw(_fp(x), _fp(y), len(x), len(y), _fp(cost), 0)
    return cost


def ref_subsequence_path(cost, starty):
    L = reference_lib()
    cost = _f32(cost)
    p = L.Path()
    ok = L.subsequence_path(_fp(cost), cost.shape[0], cost.shape[1], starty, C.byref(p))
    if not ok:
        return None, None
    px = np.ctypeslib.as_array(p.px, (p.k,)).copy()
    py = np.ctypeslib.as_array(p.py, (p.k,)).copy()
    L._libc.free(p.px)
    L._libc.free(p.py)
    return px, py


def parse_dump(path):
    """Parse the binary dump written by oracle/ref_driver.c."""
    b = open(path, "rb").read()
    o = 0

    def take(fmt):
        nonlocal o
        v = struct.unpack_from("<" + fmt, b, o)
        o += struct.calcsize("<" + fmt)
        return v

    magic, num_ref, rna, flag = take("4i")
    assert magic == 0x53464131
    names, sl, rl, so, fw, rv = [], [], [], [], [], []
    for _ in range(num_ref):
        n, seqn, off, nl = take("4i")
        names.append(b[o:o + nl].decode())
        o += nl
        fw.append(np.frombuffer(b, "<f4", n, o).copy())
        o += 4 * n
        if not rna:
            rv.append(np.frombuffer(b, "<f4", n, o).copy())
            o += 4 * n
        sl.append(seqn)
        rl.append(n)
        so.append(off)
    ref = RefSynth(names, sl, rl, so, fw, None if rna else rv)
    reads = []
    while o < len(b):
        (idl,) = take("i")
        rid = b[o:o + idl].decode()
        o += idl
        len_raw, n_ev, qs, qe = take("4q")
        (valid,) = take("b")
        rec = dict(read_id=rid, len_raw=len_raw, n_events=n_ev, qstart=qs, qend=qe, valid=bool(valid))
        if valid:
            s0, s1 = take("2Q")
            (l1,) = take("f")
            q = np.frombuffer(b, "<f4", qe - qs, o).copy()
            o += 4 * (qe - qs)
            arid, ast, aen = take("3i")
            sc, sc2 = take("2f")
            d, mq = take("bB")
            rec.update(ev_start_first=s0, ev_start_last=s1, ev_len_last=l1, query=q, rid=arid, pos_st=ast, pos_end=aen,
                       score=np.float32(sc), score2=np.float32(sc2), strand=d, mapq=mq)
        reads.append(rec)
    return dict(flag=flag, rna=bool(rna), ref=ref, reads=reads)


def reference_align_batch(events, q_off, ref, flag, threads=1, workdir="/tmp"):
    """Run the COMPILED REFERENCE's align_db (dtw_single over work_db) through oracle/_ref/ref_bench.
    Returns (rows, seconds) or None when the reference build is not available."""
    if not os.path.exists(REF_BENCH):
        return None
    ev = _f32(events)
    q_off = np.ascontiguousarray(q_off, np.int64)
    n = len(q_off) - 1
    fin = os.path.join(workdir, f"sfa_refbench_{os.getpid()}.in")
    fout = fin[:-3] + ".out"
    with open(fin, "wb") as f:
        f.write(struct.pack("<4i", flag, ref.num_ref, n, threads))
        for i in range(ref.num_ref):
            f.write(struct.pack("<3i", int(ref.ref_lengths[i]), int(ref.seq_lengths[i]), int(ref.st_offset[i])))
            f.write(_f32(ref.forward[i]).tobytes())
            if ref.reverse is not None:
                f.write(_f32(ref.reverse[i]).tobytes())
        f.write(q_off.tobytes())
        f.write(ev[:int(q_off[-1])].tobytes())
    try:
        subprocess.run([REF_BENCH, fin, fout], check=True, capture_output=True)
        b = open(fout, "rb").read()
    finally:
        for p in (fin, fout):
            if os.path.exists(p):
                os.remove(p)
    (secs,) = struct.unpack_from("<d", b, 0)
    rows = np.frombuffer(b, RESULT_DTYPE, n, 8).copy()
    return rows, secs
