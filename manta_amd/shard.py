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


class _GatherBuffers:
    """Buffers of gather_bytes, kept across calls (a bench step gathers the same amount every time): on a GPU the payload goes
    page-locked host -> device -> RCCL gather -> device -> page-locked host, each hop one DMA, no per-call allocation."""

    def __init__(self):
        self.key = None
        self.cap = 0

    def ensure(self, device, world, rank, need):
        key = (str(device), world, rank)
        if key == self.key and need <= self.cap:
            return
        self.key, self.cap = key, int(need + need // 4 + 4096)
        on_gpu = str(device).startswith("cuda")
        self.pay = torch.empty(self.cap, dtype=torch.uint8, device=device)
        self.recv = torch.empty((world, self.cap), dtype=torch.uint8, device=device) if rank == 0 else None
        self.host_pay = torch.empty(self.cap, dtype=torch.uint8, pin_memory=True) if on_gpu else None
        self.host_recv = torch.empty((world, self.cap), dtype=torch.uint8, pin_memory=True) if (on_gpu and rank == 0) else None
        # host-side fills go through numpy views: a torch CPU copy_ of megabytes starts an OpenMP team whose threads keep
        # spinning on every core afterwards (measured: the next batch call ran 2x slower in a process with default threads)
        self.fill = (self.host_pay if on_gpu else self.pay).numpy()


_BUFFERS = _GatherBuffers()


def gather_bytes(local, device="cpu", force_collectives=False):
    """one uint8 array per rank -> list of arrays on rank 0 (rank order), None elsewhere.  Only rank 0 receives the payload:
    an all_gather of the 8-byte sizes, then ONE padded gather to rank 0.  The arrays returned on a GPU run are views of a
    page-locked staging buffer that the next call overwrites."""
    local = np.ascontiguousarray(local, dtype=np.uint8)
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collectives):  # (force: tests run the collectives with one rank)
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = str(device).startswith("cuda")
    size = torch.tensor([len(local)], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, size) if hasattr(dist, "all_gather_into_tensor") and on_gpu else dist.all_gather(
        list(sizes.split(1)), size)
    sizes = [int(v) for v in sizes.cpu().tolist()]  # one device->host sync for all of them
    max_bytes = max(1, max(sizes))
    b = _BUFFERS
    b.ensure(device, world, rank, max_bytes)
    n = len(local)
    if n:
        b.fill[:n] = local
        if on_gpu:
            b.pay[:n].copy_(b.host_pay[:n], non_blocking=True)
    send = b.pay[:max_bytes]
    recv = [b.recv[r, :max_bytes] for r in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    if on_gpu:
        b.host_recv.copy_(b.recv, non_blocking=True)  # whole rows: one contiguous DMA
        torch.cuda.current_stream().synchronize()
        return [b.host_recv[r, :sizes[r]].numpy() for r in range(world)]
    return [b.recv[r, :sizes[r]].numpy().copy() for r in range(world)]


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
