"""Read-sharded multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The alignment stage shards embarrassingly over reads, so the data path has NO collective.  What travels:
  * once, at start-up: a broadcast of the reference event model (refsynth_t arrays, 239 KB for nCoV, 8 MB for a
    1 Mb reference) from rank 0 -- every rank then keeps it resident in its own HBM;
  * per batch: a gather of the fixed-size result rows (24 B/read) to rank 0, which writes PAF in read order.
The same code runs over gloo on CPU tensors (tests) and over RCCL on GPU tensors (bench.py, N > 1).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import api

# SFA_DIST_FORCE=1 runs the collectives even with a single rank (rehearsal of the RCCL path on a one-GPU box)
_FORCE = os.environ.get("SFA_DIST_FORCE") == "1"


def shard_range(n, rank, world):
    """Contiguous read range of `rank`: [rank*n/world, (rank+1)*n/world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def _dev(device):
    return torch.device(device) if device is not None else torch.device("cpu")


def broadcast_ref(ref, flag, device=None, src=0):
    """Rank `src` passes its RefModel (others pass None); every rank returns (RefModel, flag)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return ref, flag
    dev = _dev(device)
    rank = dist.get_rank()
    hdr = torch.zeros(4, dtype=torch.int64, device=dev)
    if rank == src:
        rna = ref.reverse is None
        hdr[:] = torch.tensor([ref.num_ref, int(ref.ref_lengths.astype(np.int64).sum()), int(rna), int(flag)])
    dist.broadcast(hdr, src)
    num_ref, total, rna, flag = (int(v) for v in hdr.tolist())
    meta = torch.zeros(3 * num_ref, dtype=torch.int32, device=dev)
    levels = torch.zeros(total * (1 if rna else 2), dtype=torch.float32, device=dev)
    names = [None]
    if rank == src:
        meta[:] = torch.from_numpy(np.concatenate([ref.ref_lengths, ref.st_offset, ref.seq_lengths]).astype(np.int32))
        levels[:] = torch.from_numpy(np.concatenate(ref.forward + ([] if rna else ref.reverse)))
        names = [list(ref.names)]
    dist.broadcast(meta, src)
    dist.broadcast(levels, src)
    dist.broadcast_object_list(names, src)
    if rank == src and not _FORCE:
        return ref, flag
    m = meta.cpu().numpy()
    lens, offs, seql = m[:num_ref], m[num_ref:2 * num_ref], m[2 * num_ref:]
    lv = levels.cpu().numpy()
    cuts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    fw = [lv[cuts[i]:cuts[i + 1]].copy() for i in range(num_ref)]
    rv = None if rna else [lv[total + cuts[i]:total + cuts[i + 1]].copy() for i in range(num_ref)]
    return api.RefModel(names[0], seql, lens, offs, fw, rv), flag


_PINNED = {}  # bytes -> page-locked host tensor, reused by every gather of that size


def _to_host(t):
    """Device tensor -> host numpy view of its bytes, through a cached page-locked buffer (one DMA, no further copies)."""
    if t.device.type == "cpu":
        return t.numpy()
    buf = _PINNED.get(t.numel())
    if buf is None:
        buf = _PINNED[t.numel()] = torch.empty(t.numel(), dtype=torch.uint8, pin_memory=True)
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.numpy()


def gather_rows(rows, counts, device=None, dst=0):
    """rows: this rank's result rows as a uint8 tensor [n_local*24] (GPU or CPU).  counts: reads per rank.
    Returns on `dst` a structured array with all rows in rank (= read) order, None elsewhere.  (The array may be a view
    of a reused page-locked buffer: copy it if it has to outlive the next gather of the same size.)"""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return _to_host(rows).view(api.RESULT_DTYPE)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = rows.device
    item = api.RESULT_DTYPE.itemsize
    width = max(counts) * item
    send = rows
    if rows.numel() != width:  # ragged shards: pad to the widest
        send = torch.empty(width, dtype=torch.uint8, device=dev)
        send[:rows.numel()] = rows
    big = torch.empty(world * width, dtype=torch.uint8, device=dev) if rank == dst else None
    recv = list(big.split(width)) if rank == dst else None  # views: the shards land next to each other
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    host = _to_host(big)
    if all(c * item == width for c in counts):
        return host.view(api.RESULT_DTYPE)  # equal shards: already in read order, no copy
    return np.concatenate([host[r * width:r * width + counts[r] * item] for r in range(world)]).view(api.RESULT_DTYPE)
