"""Multi-GPU sharding of candidate loci.

Loci are independent units, so the multi-GPU path has no data-path collective: every rank (one process per GPU)
takes a contiguous slice of the locus list -- the same partition the reference's `--bin-index/--bin-count` makes
(contiguous ranges balanced by cumulative cost, EdgeRetrieverBin.cpp:38-57) -- and the only communication is the
final gather of the (small, variable-length) candidate records to rank 0 over torch.distributed (RCCL on GPUs,
gloo in the CPU tests): sizes by all_gather, payload by one padded gather that only rank 0 receives.  Output order is irrelevant downstream (sortVcf.py sorts), rank order is kept anyway.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(costs, world, rank):
    """contiguous [begin, end) of items for `rank`, balancing cumulative cost like EdgeRetrieverBin"""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if n == 0:
        return 0, 0
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    edges = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world + 1)]
    edges[0], edges[-1] = 0, n
    for r in range(1, world + 1):
        edges[r] = max(edges[r], edges[r - 1])
    return edges[rank], edges[rank + 1]


def gather_bytes(local, device="cpu"):
    """one uint8 array per rank -> list of arrays on rank 0 (rank order), None elsewhere.  Only rank 0 receives the payload:
    an all_gather of the 8-byte sizes, then ONE padded gather to rank 0."""
    local = np.ascontiguousarray(local, dtype=np.uint8)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([len(local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    max_bytes = max(1, int(max(s.item() for s in sizes)))
    pay = torch.zeros(max_bytes, dtype=torch.uint8, device=device)
    if len(local):
        pay[:len(local)] = torch.from_numpy(local).to(device, non_blocking=True)
    recv = [torch.zeros_like(pay) for _ in range(world)] if rank == 0 else None
    dist.gather(pay, recv, dst=0)
    if rank != 0:
        return None
    return [recv[r][:int(sizes[r].item())].cpu().numpy() for r in range(world)]


def gather_records(local_blobs, device="cpu"):
    """local_blobs: list of bytes (one per local locus, in order).  Returns on rank 0 the concatenated list over ranks
    (rank order), elsewhere None.  The record lengths travel in front of the payload in the same gather."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_blobs)
    lens = np.array([len(b) for b in local_blobs], dtype=np.int64)
    head = np.array([len(lens)], dtype=np.int64)
    blob = np.frombuffer(head.tobytes() + lens.tobytes() + b"".join(local_blobs), dtype=np.uint8)
    got = gather_bytes(blob, device)
    if got is None:
        return None
    out = []
    for buf in got:
        n = int(np.frombuffer(buf[:8].tobytes(), dtype=np.int64)[0])
        ls = np.frombuffer(buf[8:8 + 8 * n].tobytes(), dtype=np.int64)
        raw = buf[8 + 8 * n:].tobytes()
        off = 0
        for ln in ls:
            out.append(raw[off:off + int(ln)])
            off += int(ln)
    return out
