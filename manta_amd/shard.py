"""Multi-GPU sharding of candidate loci.

Loci are independent units, so the multi-GPU path has no data-path collective: every rank (one process per GPU)
takes a contiguous slice of the locus list -- the same partition the reference's `--bin-index/--bin-count` makes
(contiguous ranges balanced by cumulative cost, EdgeRetrieverBin.cpp:38-57) -- and the only communication is the
final gather of the (small, variable-length) candidate records to rank 0 over torch.distributed (RCCL on GPUs,
gloo in the CPU tests).  Output order is irrelevant downstream (sortVcf.py sorts), rank order is kept anyway.
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


def gather_records(local_blobs, device="cpu"):
    """local_blobs: list of bytes (one per local locus, in order).  Returns on rank 0 the concatenated list over ranks
    (rank order), elsewhere None.  Two collectives: all_gather of sizes, then one padded all_gather of the payload."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_blobs)
    world, rank = dist.get_world_size(), dist.get_rank()
    lens = np.array([len(b) for b in local_blobs], dtype=np.int64)
    payload = np.frombuffer(b"".join(local_blobs), dtype=np.uint8) if len(local_blobs) else np.zeros(0, dtype=np.uint8)
    meta = torch.tensor([len(lens), len(payload)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_items = int(max(m[0].item() for m in metas))
    max_bytes = int(max(m[1].item() for m in metas))
    lens_t = torch.zeros(max(1, max_items), dtype=torch.int64, device=device)
    lens_t[:len(lens)] = torch.from_numpy(lens).to(device)
    pay_t = torch.zeros(max(1, max_bytes), dtype=torch.uint8, device=device)
    pay_t[:len(payload)] = torch.from_numpy(payload.copy()).to(device)
    all_lens = [torch.zeros_like(lens_t) for _ in range(world)]
    all_pay = [torch.zeros_like(pay_t) for _ in range(world)]
    dist.all_gather(all_lens, lens_t)
    dist.all_gather(all_pay, pay_t)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        n_items = int(metas[r][0].item())
        ls = all_lens[r][:n_items].cpu().numpy()
        buf = all_pay[r].cpu().numpy().tobytes()
        off = 0
        for ln in ls:
            out.append(buf[off:off + int(ln)])
            off += int(ln)
    return out
