"""sigfish_amd -- MI355X-native subsequence-DTW signal-to-reference alignment stage.

The product is the C-ABI shared library (include/sigfish_amd.h, built from sigfish_amd/csrc/ into
sigfish_amd/lib/libsigfish_amd.so).  This package is only a thin ctypes binding used by bench.py and the tests;
it never falls back to a CPU implementation: importing works without a GPU, creating an Aligner does not.
"""
from .api import (END, DTW, EVENT_DTYPE, INV, REF, RNA, RESULT_DTYPE, Aligner, Blow5File, RefModel, SfaError, build_id,
                  detect_events, paf_row, r2qevent_map, read_fasta, read_kmer_model, sam_row, select_query, version, znormalise)

__all__ = ["Aligner", "RefModel", "SfaError", "RESULT_DTYPE", "RNA", "DTW", "INV", "REF", "END", "paf_row",
           "read_fasta", "version", "znormalise"]
