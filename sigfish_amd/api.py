"""Host-side mirror of the reference's interface for the alignment stage, over the C-ABI.

Names follow the reference (hasindu2008/sigfish v0.2.0): a RefModel is refsynth_t (src/sigfish.h:90-99) as
produced by gen_ref (src/genref.c:86-241); Aligner.align_db is align_db (src/sigfish.c:1003-1015) for a whole
batch; result rows carry the aln_t fields (src/sigfish.h:146-158) the PAF writer (src/sigfish.c:628-660) prints.
"""
import ctypes as C
import gzip

import numpy as np

from . import _lib

# opt.flag bits (src/sigfish.h:30-39)
RNA, DTW, INV, REF, END = 0x001, 0x002, 0x004, 0x010, 0x020

RESULT_DTYPE = np.dtype([("rid", "<i4"), ("pos_st", "<i4"), ("pos_end", "<i4"), ("score", "<f4"), ("score2", "<f4"),
                         ("strand", "i1"), ("mapq", "u1"), ("valid", "u1"), ("pad", "u1")])
assert RESULT_DTYPE.itemsize == C.sizeof(_lib.SfaResult)


QUERY_INFO_DTYPE = np.dtype([("n_events", "<i8"), ("qstart", "<i8"), ("qend", "<i8"), ("start_raw_idx", "<u8"),
                             ("end_raw_idx", "<u8"), ("status", "<i4"), ("pad", "<i4")])
assert QUERY_INFO_DTYPE.itemsize == C.sizeof(_lib.SfaQueryInfo)


class SfaError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise SfaError(f"{what} failed ({rc}): {_lib.load().sfa_last_error().decode()}")


def version():
    return _lib.load().sfa_version().decode()


def build_id():
    return _lib.load().sfa_build_id().decode()


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def znormalise(v):
    v = _f32(v).copy()
    _lib.load().sfa_znormalise(v.ctypes.data_as(_lib.f32p), len(v))
    return v


def read_fasta(path):
    """[(name, sequence)]: name is the first word of the header, sequence lines are concatenated (kseq.h)."""
    op = gzip.open if str(path).endswith(".gz") else open
    recs, name, chunks = [], None, []
    with op(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(chunks)))
                w = line[1:].split()
                name, chunks = (w[0] if w else ""), []
            elif name is not None:
                chunks.append("".join(line.split()))
    if name is not None:
        recs.append((name, "".join(chunks)))
    return recs


class RefModel:
    """refsynth_t: per-contig z-normalised expected event levels, forward and (DNA) reverse complement."""

    def __init__(self, names, seq_lengths, ref_lengths, st_offset, forward, reverse):
        self.names = list(names)
        self.seq_lengths = np.asarray(seq_lengths, np.int32)
        self.ref_lengths = np.ascontiguousarray(ref_lengths, np.int32)
        self.st_offset = np.ascontiguousarray(st_offset, np.int32)
        self.forward = [_f32(a) for a in forward]
        self.reverse = None if reverse is None else [_f32(a) for a in reverse]
        self.num_ref = len(self.names)

    @classmethod
    def from_records(cls, records, level_mean, k, flag=0, query_size=250):
        """gen_ref (src/genref.c:86-241) over [(name, sequence)] with a 4^k table of k-mer level means."""
        L = _lib.load()
        lv = _f32(level_mean)
        if len(lv) != 4 ** k:
            raise ValueError(f"k-mer model needs {4 ** k} levels, got {len(lv)}")
        rna = bool(flag & RNA)
        names, sl, rl, so, fw, rv = [], [], [], [], [], []
        for name, seq in records:
            b = seq.encode()
            cap = max(len(b) + 1 - k, 1)
            f = np.zeros(cap, np.float32)
            r = np.zeros(cap, np.float32)
            off = C.c_int32(0)
            n = L.sfa_gen_ref_record(b, len(b), lv.ctypes.data_as(_lib.f32p), k, flag, query_size,
                                     f.ctypes.data_as(_lib.f32p), None if rna else r.ctypes.data_as(_lib.f32p),
                                     C.byref(off))
            if n <= 0:
                raise SfaError(f"contig {name}: cannot build reference events (length {len(b)}, k={k})")
            names.append(name)
            sl.append(len(b))
            rl.append(n)
            so.append(off.value)
            fw.append(f[:n].copy())
            rv.append(r[:n].copy())
        return cls(names, sl, rl, so, fw, None if rna else rv)

    @classmethod
    def from_fasta(cls, path, level_mean, k, flag=0, query_size=250):
        return cls.from_records(read_fasta(path), level_mean, k, flag, query_size)

    def total_columns(self):
        return int(self.ref_lengths.sum()) * (1 if self.reverse is None else 2)

    def _as_c(self):
        n = self.num_ref
        fa = (_lib.f32p * n)(*[a.ctypes.data_as(_lib.f32p) for a in self.forward])
        ra = None
        if self.reverse is not None:
            ra = (_lib.f32p * n)(*[a.ctypes.data_as(_lib.f32p) for a in self.reverse])
        ref = _lib.SfaRef(n, self.ref_lengths.ctypes.data_as(_lib.i32p), self.st_offset.ctypes.data_as(_lib.i32p),
                          fa, C.cast(ra, C.POINTER(_lib.f32p)) if ra is not None else None)
        return ref, (fa, ra)


class Aligner:
    """The accelerator context: reference arrays resident in HBM, batches aligned by the gfx950 kernels."""

    def __init__(self, ref: RefModel, flag=0, device=0, devices=None):
        """device: one GPU (sfa_init).  devices: a list of GPUs (sfa_init_devices) -- every batch is then sharded over them
        in contiguous read ranges and comes back in input order; a GPU may be listed more than once."""
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.ref, self.flag, self.device = ref, int(flag), int(device)
        cref, keep = ref._as_c()
        if devices is None:
            _check(self._L.sfa_init(C.byref(self._h), C.byref(cref), self.flag, self.device), "sfa_init")
        else:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(self._L.sfa_init_devices(C.byref(self._h), C.byref(cref), self.flag, devs, len(devices)), "sfa_init_devices")
        del keep

    def n_devices(self):
        return int(self._L.sfa_n_devices(self._h))

    def close(self):
        if self._h:
            self._L.sfa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- align_db over packed host arrays -------------------------------------------------------------------
    def align_db(self, queries, q_off):
        """queries: concatenated z-normalised event means (event order); q_off: int64[n+1]. -> RESULT_DTYPE[n]"""
        q = _f32(queries)
        qo = np.ascontiguousarray(q_off, np.int64)
        n = len(qo) - 1
        out = np.zeros(n, RESULT_DTYPE)
        if q.size == 0:
            q = np.zeros(1, np.float32)
        _check(self._L.sfa_align_batch(self._h, q.ctypes.data_as(_lib.f32p), qo.ctypes.data_as(_lib.i64p), n,
                                       out.ctypes.data_as(C.c_void_p)), "sfa_align_batch")
        return out

    def submit(self, queries, q_off):
        """First half of align_db: queue the batch and return; collect the rows with wait().  One batch in flight."""
        q = _f32(queries)
        qo = np.ascontiguousarray(q_off, np.int64)
        if q.size == 0:
            q = np.zeros(1, np.float32)
        self._pending = (q, qo)  # the library reads `queries` until wait()
        _check(self._L.sfa_submit_batch(self._h, q.ctypes.data_as(_lib.f32p), qo.ctypes.data_as(_lib.i64p), len(qo) - 1),
               "sfa_submit_batch")

    def wait(self):
        q, qo = getattr(self, "_pending", None) or (None, np.zeros(1, np.int64))
        n = len(qo) - 1
        out = np.zeros(n, RESULT_DTYPE)
        try:
            _check(self._L.sfa_wait_batch(self._h, out.ctypes.data_as(C.c_void_p), n), "sfa_wait_batch")
        finally:
            self._pending = None
        return out

    # -- same with device-resident buffers (torch tensors or raw pointers) ------------------------------------
    def align_db_device(self, d_queries_ptr, q_off, n, d_out_ptr, sync=True):
        qo = np.ascontiguousarray(q_off, np.int64)
        _check(self._L.sfa_align_batch_device(self._h, C.c_void_p(d_queries_ptr), qo.ctypes.data_as(_lib.i64p), n,
                                              C.c_void_p(d_out_ptr), 1 if sync else 0), "sfa_align_batch_device")

    def align_events(self, event_tables, qstart, qend):
        """event_tables: list of structured arrays with sfa_event_t layout (or None)."""
        n = len(event_tables)
        EP = C.POINTER(_lib.SfaEvent)
        ptrs = (EP * n)()
        nev = np.zeros(n, np.int64)
        keep = []
        for i, t in enumerate(event_tables):
            if t is None or len(t) == 0:
                ptrs[i] = None
                continue
            a = np.ascontiguousarray(t)
            keep.append(a)
            ptrs[i] = C.cast(a.ctypes.data, EP)
            nev[i] = len(a)
        qs = np.ascontiguousarray(qstart, np.int64)
        qe = np.ascontiguousarray(qend, np.int64)
        out = np.zeros(n, RESULT_DTYPE)
        _check(self._L.sfa_align_events(self._h, ptrs, nev.ctypes.data_as(_lib.i64p), qs.ctypes.data_as(_lib.i64p),
                                        qe.ctypes.data_as(_lib.i64p), n, out.ctypes.data_as(C.c_void_p)),
               "sfa_align_events")
        return out

    def set_option(self, key, value):
        _check(self._L.sfa_set_option(self._h, key.encode(), int(value)), f"sfa_set_option({key})")

    def align_raw(self, raw, raw_off, scaling, prefix_size=50, query_size=250, return_events=False):
        """process_db on the device: raw int16 samples (concatenated) -> (rows, info).  scaling: float64 [n,3] =
        digitisation, offset, range per read.  return_events: also the query windows' event tables,
        EVENT_DTYPE[n, query_size] (means z-normalised; read i uses the first info.qend - info.qstart entries)."""
        raw = np.ascontiguousarray(raw, np.int16)
        ro = np.ascontiguousarray(raw_off, np.int64)
        sc = np.ascontiguousarray(scaling, np.float64).reshape(-1)
        n = len(ro) - 1
        rows = np.zeros(n, RESULT_DTYPE)
        info = np.zeros(n, QUERY_INFO_DTYPE)
        if raw.size == 0:
            raw = np.zeros(1, np.int16)
        qev = np.zeros((n, query_size), EVENT_DTYPE) if return_events else None
        _check(self._L.sfa_align_raw_ex(self._h, raw.ctypes.data_as(C.POINTER(C.c_int16)), ro.ctypes.data_as(_lib.i64p),
                                        sc.ctypes.data_as(C.POINTER(C.c_double)), n, prefix_size, query_size,
                                        rows.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p),
                                        qev.ctypes.data_as(C.c_void_p) if return_events else None), "sfa_align_raw_ex")
        return (rows, info, qev) if return_events else (rows, info)

    def align_blow5(self, records, rec_off, record_zlib, signal_svb, prefix_size=50, query_size=250, return_events=False):
        """load_db's records straight to the device: `records` = the BLOW5 records of a batch back to back (bytes / uint8 array,
        without their size prefixes), rec_off int64[n+1].  -> (rows, info, heads[, query events])"""
        rec = np.frombuffer(records, np.uint8) if isinstance(records, (bytes, bytearray, memoryview)) else np.ascontiguousarray(records, np.uint8)
        ro = np.ascontiguousarray(rec_off, np.int64)
        n = len(ro) - 1
        rows = np.zeros(n, RESULT_DTYPE)
        info = np.zeros(n, QUERY_INFO_DTYPE)
        heads = (_lib.SfaReadHead * max(n, 1))()
        qev = np.zeros((n, query_size), EVENT_DTYPE) if return_events else None
        if rec.size == 0:
            rec = np.zeros(1, np.uint8)
        _check(self._L.sfa_align_blow5(self._h, rec.ctypes.data_as(C.c_void_p), ro.ctypes.data_as(_lib.i64p), n, int(bool(record_zlib)),
                                       int(bool(signal_svb)), prefix_size, query_size, rows.ctypes.data_as(C.c_void_p),
                                       info.ctypes.data_as(C.c_void_p), C.cast(heads, C.c_void_p),
                                       qev.ctypes.data_as(C.c_void_p) if return_events else None), "sfa_align_blow5")
        hs = [dict(read_id=heads[i].read_id.decode(), n_samples=heads[i].n_samples, digitisation=heads[i].digitisation,
                   offset=heads[i].offset, range=heads[i].range, record_bytes=heads[i].record_bytes) for i in range(n)]
        return (rows, info, hs, qev) if return_events else (rows, info, hs)

    def inflate_device(self, streams, cap_factor=4, cap_extra=4096):
        """The device-side DEFLATE decoder alone: list of zlib streams -> list of bytes (None where the decoder declined)."""
        n = len(streams)
        in_off = np.concatenate([[0], np.cumsum([len(x) for x in streams])]).astype(np.int64)
        blob = np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8)
        out_off = np.concatenate([[0], np.cumsum([len(x) * cap_factor + cap_extra for x in streams])]).astype(np.int64)
        out = np.zeros(int(out_off[-1]) + 8, np.uint8)
        lens = np.zeros(max(n, 1), np.int32)
        _check(self._L.sfa_inflate_zlib_device(self._h, blob.ctypes.data_as(C.c_void_p), in_off.ctypes.data_as(_lib.i64p), n,
                                               out.ctypes.data_as(C.c_void_p), out_off.ctypes.data_as(_lib.i64p),
                                               lens.ctypes.data_as(_lib.i32p)), "sfa_inflate_zlib_device")
        return [None if lens[i] < 0 else out[out_off[i]:out_off[i] + lens[i]].tobytes() for i in range(n)]

    def sync(self):
        _check(self._L.sfa_sync(self._h), "sfa_sync")

    def profile(self):
        p = _lib.SfaProfile()
        _check(self._L.sfa_get_profile(self._h, C.byref(p)), "sfa_get_profile")
        return {k: getattr(p, k) for k, _ in _lib.SfaProfile._fields_}

    def stream(self):
        return self._L.sfa_stream(self._h)


def plan_batch(q_off, job_len, ckpt_interval=0, ckpt_budget_bytes=0, lane_widening=0):
    """Host-side batch layout (no GPU needed): (info dict, slot_of_read int32[n])."""
    qo = np.ascontiguousarray(q_off, np.int64)
    jl = np.ascontiguousarray(job_len, np.int32)
    n = len(qo) - 1
    slot = np.zeros(max(n, 1), np.int32)
    info = _lib.SfaPlanInfo()
    _check(_lib.load().sfa_plan_batch(qo.ctypes.data_as(_lib.i64p), n, jl.ctypes.data_as(_lib.i32p), len(jl),
                                      int(ckpt_interval), int(ckpt_budget_bytes), int(lane_widening),
                                      slot.ctypes.data_as(_lib.i32p),
                                      C.byref(info)), "sfa_plan_batch")
    return {k: getattr(info, k) for k, _ in _lib.SfaPlanInfo._fields_}, slot[:n]


EVENT_DTYPE = np.dtype([("start", "<u8"), ("length", "<f4"), ("mean", "<f4"), ("stdv", "<f4")], align=True)


def paf_row(res, read_id, rname, start_raw, end_raw, query_size, len_raw, rlength):
    """paf_str (src/sigfish.c:628-660) for one result row."""
    r = _lib.SfaResult(int(res["rid"]), int(res["pos_st"]), int(res["pos_end"]), float(res["score"]),
                       float(res["score2"]), int(res["strand"]), int(res["mapq"]), int(res["valid"]), 0)
    buf = C.create_string_buffer(4096)
    n = _lib.load().sfa_paf_row(buf, 4096, C.byref(r), str(read_id).encode(), str(rname).encode(), int(start_raw),
                                int(end_raw), int(query_size), int(len_raw), int(rlength))
    if n < 0:
        raise SfaError("sfa_paf_row: buffer too small")
    return buf.raw[:n].decode()


# ---- host pre-DP stages and readers ------------------------------------------------------------
class Blow5File:
    """Sequential BLOW5 reader: iterates (read_id, meta dict, raw int16 array)."""

    def __init__(self, path):
        self._L = _lib.load()
        self._h = self._L.sfa_blow5_open(str(path).encode())
        if not self._h:
            raise SfaError(self._L.sfa_last_error().decode())

    def attr(self, key):
        v = self._L.sfa_blow5_attr(self._h, key.encode())
        return None if v is None else v.decode()

    def select_shard(self, r, G):
        """Only the records starting in the r-th of G equal byte slices of the file (one rank of a read-sharded run)."""
        if self._L.sfa_blow5_select_shard(self._h, r, G) != 0:
            raise SfaError(self._L.sfa_last_error().decode())
        return self

    def select_records(self, first, count=-1):
        """Only records [first, first + count) by position in the file (count < 0: to the end)."""
        if self._L.sfa_blow5_select_records(self._h, first, count) != 0:
            raise SfaError(self._L.sfa_last_error().decode())
        return self

    def __iter__(self):
        rid = C.c_char_p()
        meta = (C.c_double * 4)()
        raw = C.POINTER(C.c_int16)()
        n = C.c_int64()
        while True:
            rc = self._L.sfa_blow5_next(self._h, C.byref(rid), meta, C.byref(raw), C.byref(n))
            if rc == 0:
                return
            if rc < 0:
                raise SfaError(self._L.sfa_last_error().decode())
            sig = np.ctypeslib.as_array(raw, (n.value,)).copy() if n.value else np.zeros(0, np.int16)
            yield rid.value.decode(), dict(digitisation=meta[0], offset=meta[1], range=meta[2], sampling_rate=meta[3]), sig

    def close(self):
        if self._h:
            self._L.sfa_blow5_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def detect_events(raw, meta, rna=False):
    """event_single: raw int16 samples + scaling -> EVENT_DTYPE array."""
    raw = np.ascontiguousarray(raw, np.int16)
    L = _lib.load()
    cap = max(len(raw) // 2, 16)
    out = np.zeros(cap, EVENT_DTYPE)
    n = L.sfa_detect_events(raw.ctypes.data_as(C.POINTER(C.c_int16)), len(raw), meta["digitisation"], meta["offset"],
                            meta["range"], int(rna), C.cast(out.ctypes.data, C.POINTER(_lib.SfaEvent)), cap)
    if n < 0:
        raise SfaError("sfa_detect_events failed")
    if n > cap:
        out = np.zeros(n, EVENT_DTYPE)
        n = L.sfa_detect_events(raw.ctypes.data_as(C.POINTER(C.c_int16)), len(raw), meta["digitisation"], meta["offset"],
                                meta["range"], int(rna), C.cast(out.ctypes.data, C.POINTER(_lib.SfaEvent)), n)
    return out[:n]


def select_query(events, raw, meta, prefix_size=50, query_size=250, flag=0, pore=0):
    """normalise_single: returns (keep, qstart, qend); `events` is normalised in place over [qstart,qend)."""
    raw = np.ascontiguousarray(raw, np.int16)
    qs, qe = C.c_int64(), C.c_int64()
    keep = _lib.load().sfa_select_query(C.cast(events.ctypes.data, C.POINTER(_lib.SfaEvent)), len(events),
                                        raw.ctypes.data_as(C.POINTER(C.c_int16)), len(raw), meta["digitisation"],
                                        meta["offset"], meta["range"], prefix_size, query_size, flag, pore,
                                        C.byref(qs), C.byref(qe))
    return bool(keep), qs.value, qe.value


def read_kmer_model(path, warnings=None):
    """read_model: (level_mean[4^k], k).  Rows the reference would only log as corrupted are appended to `warnings` (a list)."""
    lv = np.zeros(262144, np.float32)
    k = C.c_uint32()
    L = _lib.load()
    _check(L.sfa_read_kmer_model(str(path).encode(), lv.ctypes.data_as(_lib.f32p), C.byref(k)), "sfa_read_kmer_model")
    if warnings is not None:
        warnings.extend(w for w in L.sfa_last_error().decode().split("\n") if w)
    return lv[:4 ** k.value].copy(), k.value


def r2qevent_map(res, events, qstart, qend, ref_array, ref_st_offset, flag):
    """aln_t.r2qevent_map (path_to_map, src/sigfish.c:530-571) for one result row: int32 [r2qevent_size, 2] = start, stop."""
    r = _lib.SfaResult(int(res["rid"]), int(res["pos_st"]), int(res["pos_end"]), float(res["score"]),
                       float(res["score2"]), int(res["strand"]), int(res["mapq"]), int(res["valid"]), 0)
    y = _f32(ref_array)
    L = _lib.load()
    evp = C.cast(events.ctypes.data, C.POINTER(_lib.SfaEvent))
    need = L.sfa_r2qevent_map(C.byref(r), evp, int(qstart), int(qend), y.ctypes.data_as(_lib.f32p), len(y), int(ref_st_offset),
                              int(flag), None, 0)
    if need < 0:
        raise SfaError(f"sfa_r2qevent_map failed ({need})")
    out = np.zeros((need, 2), np.int32)
    n = L.sfa_r2qevent_map(C.byref(r), evp, int(qstart), int(qend), y.ctypes.data_as(_lib.f32p), len(y), int(ref_st_offset),
                           int(flag), out.ctypes.data_as(_lib.i32p), need)
    if n != need:
        raise SfaError(f"sfa_r2qevent_map failed ({n})")
    return out


def sam_row(res, read_id, rname, events, qstart, qend, ref_array, ref_st_offset, flag):
    """sam_str (src/sigfish.c:770-794) for one result row; `events` normalised as for align_events."""
    r = _lib.SfaResult(int(res["rid"]), int(res["pos_st"]), int(res["pos_end"]), float(res["score"]),
                       float(res["score2"]), int(res["strand"]), int(res["mapq"]), int(res["valid"]), 0)
    y = _f32(ref_array)
    buf = C.create_string_buffer(1 << 22)
    n = _lib.load().sfa_sam_row(buf, len(buf), C.byref(r), str(read_id).encode(), str(rname).encode(),
                                C.cast(events.ctypes.data, C.POINTER(_lib.SfaEvent)), int(qstart), int(qend),
                                y.ctypes.data_as(_lib.f32p), len(y), int(ref_st_offset), int(flag))
    if n < 0:
        raise SfaError(f"sfa_sam_row failed ({n})")
    return buf.raw[:n].decode()
