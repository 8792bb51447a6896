"""The N>1 path on CPU: world_size 2 over gloo.  Each rank receives the reference model by broadcast, aligns
its contiguous shard of reads (the oracle stands in for the GPU here -- this test is about the sharding, the
broadcast and the ordered gather, not about the kernels) and rank 0 must end up with exactly the rows a single
process produces."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_reads, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sigfish_amd as S
    from oracle import oracle as O
    from sigfish_amd import dist as D
    from sigfish_amd import synth

    ref = flag = None
    if rank == 0:
        ref, flag, _, _, _ = synth.workload("sequin_r9_rna_q250", n_reads=4, seed=0)
    ref, flag = D.broadcast_ref(ref, flag)
    assert ref.reverse is None and ref.num_ref == 160 and flag == S.RNA
    # every rank regenerates the same global batch and takes its contiguous range
    q, q_off, _ = synth.make_reads(ref, n_reads, qlen=250, seed=42)
    lo, hi = D.shard_range(n_reads, rank, world)
    oref = O.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    rows = O.align_batch(q[q_off[lo]:q_off[hi]], q_off[lo:hi + 1] - q_off[lo], oref, flag, threads=2)
    counts = [D.shard_range(n_reads, r, world)[1] - D.shard_range(n_reads, r, world)[0] for r in range(world)]
    allrows = D.gather_rows(torch.from_numpy(np.frombuffer(rows.tobytes(), np.uint8).copy()), counts)
    if rank == 0:
        np.save(out_path, allrows)
    else:
        assert allrows is None
    dist.barrier()
    dist.destroy_process_group()


def test_read_sharding_world2(tmp_path, oracle):
    import sigfish_amd as S
    from sigfish_amd import synth
    n = 11  # odd on purpose: shards of 5 and 6 reads
    out = str(tmp_path / "rows.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    ref, flag, _, _, _ = synth.workload("sequin_r9_rna_q250", n_reads=4, seed=0)
    q, q_off, _ = synth.make_reads(ref, n, qlen=250, seed=42)
    oref = oracle.RefSynth(ref.names, ref.seq_lengths, ref.ref_lengths, ref.st_offset, ref.forward, ref.reverse)
    want = oracle.align_batch(q, q_off, oref, flag, threads=4)
    assert got.dtype == S.RESULT_DTYPE and got.tobytes() == want.tobytes()


def test_shard_ranges_cover_everything():
    from sigfish_amd.dist import shard_range
    for n in (0, 1, 7, 8, 100_000, 1_000_003):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
