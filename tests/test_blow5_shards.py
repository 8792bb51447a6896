"""Read sharding of a BLOW5 file (contiguous read ranges per rank): the G byte shards of a file are every record
exactly once and in file order, whatever G; record ranges select by position; a selection past the end is empty.  CPU only --
this is the reader every rank of `sigfish-amd dtw --ranks G` opens (host/blow5.cpp: select_shard / select_records)."""
import os

import numpy as np
import pytest

import sigfish_amd as S
from tests.util import GOLD, write_blow5

FILES = [os.path.join(GOLD, "data", "sp1_dna.blow5"), os.path.join(GOLD, "data", "sequin_rna.blow5"),
         os.path.join(GOLD, "random", "rnd_dna.blow5")]


def _all(f):
    return [(rid, meta["digitisation"], meta["offset"], meta["range"], raw.tobytes()) for rid, meta, raw in f]


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_shards_partition_the_file_in_order(path):
    whole = _all(S.Blow5File(path))
    assert whole
    for G in (1, 2, 3, 4, 7, 8, len(whole), len(whole) + 5, 64):
        parts = [_all(S.Blow5File(path).select_shard(r, G)) for r in range(G)]
        assert sum(parts, []) == whole, G
        if G <= len(whole) // 2:  # byte-balanced: with far fewer ranks than records nobody goes empty
            assert all(parts), (G, [len(p) for p in parts])


def test_shards_of_ragged_records(tmp_path):
    """Record sizes from 2 bytes of signal to 40 000: a shard boundary falls inside a record far more often than on one."""
    rng = np.random.default_rng(5)
    reads = [(f"r{i}", 8192.0, 3.0, 1400.0, 4000.0, rng.integers(-500, 500, int(n)).astype(np.int16))
             for i, n in enumerate(rng.choice([1, 7, 300, 4000, 20000], 60))]
    for compress in (False, True):
        p = str(tmp_path / f"ragged{int(compress)}.blow5")
        write_blow5(p, reads, compress=compress)
        whole = _all(S.Blow5File(p))
        assert [w[0] for w in whole] == [r[0] for r in reads]
        for G in (2, 3, 5, 16, 61):
            assert sum((_all(S.Blow5File(p).select_shard(r, G)) for r in range(G)), []) == whole, (compress, G)


def test_record_ranges(tmp_path):
    path = FILES[2]
    whole = _all(S.Blow5File(path))
    n = len(whole)
    assert _all(S.Blow5File(path).select_records(0)) == whole
    assert _all(S.Blow5File(path).select_records(3, 4)) == whole[3:7]
    assert _all(S.Blow5File(path).select_records(n - 2)) == whole[n - 2:]
    assert _all(S.Blow5File(path).select_records(n - 2, 100)) == whole[n - 2:]
    assert _all(S.Blow5File(path).select_records(5, 0)) == []
    assert _all(S.Blow5File(path).select_records(n)) == []
    assert _all(S.Blow5File(path).select_records(n + 10, 3)) == []   # beyond the end: empty, not an error


def test_bad_selections_and_empty_files(tmp_path):
    f = S.Blow5File(FILES[0])
    with pytest.raises(S.SfaError):
        f.select_shard(2, 2)
    with pytest.raises(S.SfaError):
        f.select_shard(0, 0)
    with pytest.raises(S.SfaError):
        f.select_records(-1)
    empty = str(tmp_path / "empty.blow5")
    write_blow5(empty, [])
    for G in (1, 3):
        assert [_all(S.Blow5File(empty).select_shard(r, G)) for r in range(G)] == [[]] * G
    # a file cut inside a record: the walk to a later shard reports it instead of running off the mapping
    data = open(FILES[0], "rb").read()
    cut = str(tmp_path / "cut.blow5")
    open(cut, "wb").write(data[:len(data) // 2])
    with pytest.raises(S.SfaError):
        _all(S.Blow5File(cut).select_shard(3, 4))
