"""ctypes loader for sigfish_amd/lib/libsigfish_amd.so (the C-ABI in include/sigfish_amd.h)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFA_LIB") or os.path.join(HERE, "lib", "libsigfish_amd.so")  # SFA_LIB: A/B builds

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class SfaRef(C.Structure):
    _fields_ = [("num_ref", C.c_int32), ("ref_lengths", i32p), ("ref_st_offset", i32p),
                ("forward", C.POINTER(f32p)), ("reverse", C.POINTER(f32p))]


class SfaResult(C.Structure):
    _fields_ = [("rid", C.c_int32), ("pos_st", C.c_int32), ("pos_end", C.c_int32), ("score", C.c_float),
                ("score2", C.c_float), ("strand", C.c_int8), ("mapq", C.c_uint8), ("valid", C.c_uint8),
                ("pad", C.c_uint8)]


class SfaProfile(C.Structure):
    _fields_ = [("fill_ms", C.c_double), ("trace_ms", C.c_double), ("finalize_ms", C.c_double),
                ("total_ms", C.c_double), ("cells", C.c_int64), ("fill_launches", C.c_int64),
                ("ckpt_interval", C.c_int64), ("ckpt_bytes", C.c_int64), ("n_tasks", C.c_int64),
                ("n_chunks", C.c_int64), ("n_segments", C.c_int64), ("segment_reruns", C.c_int64),
                ("events_ms", C.c_double), ("normalise_ms", C.c_double), ("non_finite_reads", C.c_int64), ("decode_ms", C.c_double), ("blow5_fallbacks", C.c_int64), ("lds_ckpt", C.c_int64), ("trace_margin", C.c_int64), ("fused_trace", C.c_int64)]


class SfaPlanInfo(C.Structure):
    _fields_ = [("n_quads", C.c_int32), ("n_chunks", C.c_int32), ("n_classes", C.c_int32),
                ("max_rows_per_lane", C.c_int32), ("ckpt_interval", C.c_int32), ("trace_margin", C.c_int32),
                ("ckpt_bytes", C.c_int64), ("n_tasks", C.c_int64), ("max_lanes_per_read", C.c_int32),
                ("lane_widening", C.c_int32)]


class SfaQueryInfo(C.Structure):
    _fields_ = [("n_events", C.c_int64), ("qstart", C.c_int64), ("qend", C.c_int64), ("start_raw_idx", C.c_uint64),
                ("end_raw_idx", C.c_uint64), ("status", C.c_int32), ("pad", C.c_int32)]


class SfaReadHead(C.Structure):
    _fields_ = [("read_id", C.c_char * 128), ("id_len", C.c_int32), ("pad", C.c_int32), ("n_samples", C.c_int64),
                ("digitisation", C.c_double), ("offset", C.c_double), ("range", C.c_double), ("record_bytes", C.c_int64)]


class SfaEvent(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_float), ("mean", C.c_float), ("stdv", C.c_float)]


# every symbol include/sigfish_amd.h declares (checked by tests/test_capi_host.py::test_library_exports_every_declared_symbol)
SYMBOLS = ["sfa_init", "sfa_init_devices", "sfa_n_devices", "sfa_align_batch", "sfa_submit_batch", "sfa_wait_batch", "sfa_align_batch_device", "sfa_align_events", "sfa_align_raw", "sfa_align_raw_ex", "sfa_align_blow5", "sfa_inflate_zlib_device", "sfa_pinned_alloc", "sfa_pinned_free", "sfa_sync",
           "sfa_get_profile", "sfa_stream", "sfa_set_option", "sfa_plan_batch", "sfa_destroy", "sfa_last_error", "sfa_version", "sfa_build_id", "sfa_gen_ref_record",
           "sfa_znormalise", "sfa_paf_row", "sfa_sam_row", "sfa_r2qevent_map", "sfa_detect_events", "sfa_select_query", "sfa_read_kmer_model",
           "sfa_blow5_open", "sfa_blow5_attr", "sfa_blow5_next", "sfa_blow5_close", "sfa_blow5_select_shard", "sfa_blow5_select_records", "sfa_inflate_zlib", "sfa_inflate_zlib_pair", "sfa_device_memory"]

_lib = None


def load():
    """Load the native library; fails loudly when it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C sigfish_amd/csrc` "
                          "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sfa_init.argtypes = [C.POINTER(vp), C.POINTER(SfaRef), C.c_uint32, C.c_int]
    L.sfa_init_devices.argtypes = [C.POINTER(vp), C.POINTER(SfaRef), C.c_uint32, C.POINTER(C.c_int), C.c_int]
    L.sfa_n_devices.argtypes = [vp]
    L.sfa_align_batch.argtypes = [vp, f32p, i64p, C.c_int32, vp]
    L.sfa_submit_batch.argtypes = [vp, f32p, i64p, C.c_int32]
    L.sfa_wait_batch.argtypes = [vp, vp, C.c_int32]
    L.sfa_align_batch_device.argtypes = [vp, vp, i64p, C.c_int32, vp, C.c_int]
    L.sfa_align_events.argtypes = [vp, C.POINTER(C.POINTER(SfaEvent)), i64p, i64p, i64p, C.c_int32, vp]
    L.sfa_align_raw.argtypes = [vp, C.POINTER(C.c_int16), i64p, C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_int32, vp, vp]
    L.sfa_align_raw_ex.argtypes = [vp, C.POINTER(C.c_int16), i64p, C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    L.sfa_align_blow5.argtypes = [vp, vp, i64p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
    L.sfa_inflate_zlib_device.argtypes = [vp, vp, i64p, C.c_int32, vp, i64p, i32p]
    L.sfa_pinned_alloc.argtypes = [C.c_size_t]
    L.sfa_pinned_alloc.restype = vp
    L.sfa_pinned_free.argtypes = [vp]
    L.sfa_pinned_free.restype = None
    L.sfa_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.sfa_plan_batch.argtypes = [i64p, C.c_int32, i32p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, i32p, C.POINTER(SfaPlanInfo)]
    L.sfa_sync.argtypes = [vp]
    L.sfa_get_profile.argtypes = [vp, C.POINTER(SfaProfile)]
    L.sfa_stream.argtypes = [vp]
    L.sfa_stream.restype = vp
    L.sfa_destroy.argtypes = [vp]
    L.sfa_destroy.restype = None
    L.sfa_last_error.restype = C.c_char_p
    L.sfa_version.restype = C.c_char_p
    L.sfa_build_id.restype = C.c_char_p
    L.sfa_gen_ref_record.argtypes = [C.c_char_p, C.c_int32, f32p, C.c_uint32, C.c_uint32, C.c_int32, f32p, f32p,
                                     i32p]
    L.sfa_gen_ref_record.restype = C.c_int32
    L.sfa_znormalise.argtypes = [f32p, C.c_uint64]
    L.sfa_znormalise.restype = None
    L.sfa_paf_row.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(SfaResult), C.c_char_p, C.c_char_p, C.c_uint64,
                              C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    i16p = C.POINTER(C.c_int16)
    L.sfa_sam_row.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(SfaResult), C.c_char_p, C.c_char_p, C.POINTER(SfaEvent),
                              C.c_int64, C.c_int64, f32p, C.c_int32, C.c_int32, C.c_uint32]
    L.sfa_r2qevent_map.argtypes = [C.POINTER(SfaResult), C.POINTER(SfaEvent), C.c_int64, C.c_int64, f32p, C.c_int32, C.c_int32,
                                   C.c_uint32, i32p, C.c_int32]
    L.sfa_r2qevent_map.restype = C.c_int32
    L.sfa_detect_events.argtypes = [i16p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(SfaEvent),
                                    C.c_int64]
    L.sfa_detect_events.restype = C.c_int64
    L.sfa_select_query.argtypes = [C.POINTER(SfaEvent), C.c_int64, i16p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                   C.c_int32, C.c_int32, C.c_uint32, C.c_int, i64p, i64p]
    L.sfa_read_kmer_model.argtypes = [C.c_char_p, f32p, C.POINTER(C.c_uint32)]
    L.sfa_blow5_open.argtypes = [C.c_char_p]
    L.sfa_blow5_open.restype = vp
    L.sfa_blow5_attr.argtypes = [vp, C.c_char_p]
    L.sfa_blow5_attr.restype = C.c_char_p
    L.sfa_blow5_next.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(i16p), i64p]
    L.sfa_blow5_close.argtypes = [vp]
    L.sfa_blow5_close.restype = None
    L.sfa_blow5_select_shard.argtypes = [vp, C.c_int32, C.c_int32]
    L.sfa_blow5_select_records.argtypes = [vp, C.c_int64, C.c_int64]
    L.sfa_inflate_zlib.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_size_t]
    L.sfa_inflate_zlib_pair.restype = C.c_int
    L.sfa_inflate_zlib_pair.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, i64p]
    L.sfa_inflate_zlib.restype = C.c_int64
    L.sfa_device_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _lib = L
    return L
