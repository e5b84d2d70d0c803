"""Multi-GPU batched mode: shard documents across ranks, gather per-shard counts.

The path shards by DOCUMENT (independent units): rank r gets a contiguous, byte-balanced range of
documents, runs the same single-GPU kernels on it, and keeps its outputs local.  The only collective is one
all_gather of 4 x int64 per rank -- {documents, structurals, string bytes, failed documents} -- so that every rank
knows the global output offsets (north_star: "RCCL over xGMI only as a gather of per-shard counts").  On GPUs the
process group is "nccl" (= RCCL); the CPU tests run the same code over "gloo".
"""
import numpy as np


def partition_documents(doc_offsets, world_size):
    """Contiguous, byte-balanced document ranges: -> list of (first_doc, last_doc_exclusive) per rank.
    doc_offsets has n+1 entries (doc k = [offsets[k], offsets[k+1]))."""
    offs = np.asarray(doc_offsets, dtype=np.uint64)
    n = offs.size - 1
    total = int(offs[-1] - offs[0])
    bounds = [0]
    for r in range(1, world_size):
        target = int(offs[0]) + (total * r) // world_size
        k = int(np.searchsorted(offs, np.uint64(target), side="left"))
        k = min(max(k, bounds[-1]), n)
        bounds.append(k)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def gather_counts(local_counts, device=None):
    """all_gather of the per-shard {docs, structurals, string_bytes, failed_docs}; -> int64 tensor [world, 4].
    With a single process (no initialised process group) returns the local row."""
    import torch
    import torch.distributed as dist
    row = torch.tensor([int(x) for x in local_counts], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return row[None, :]
    out = torch.empty(dist.get_world_size() * row.numel(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, row)
    return out.view(dist.get_world_size(), row.numel())


def global_offsets(gathered):
    """Exclusive prefix over ranks of the gathered counts: row r = where rank r's outputs start globally."""
    import torch
    g = gathered.to(torch.int64)
    return torch.cumsum(g, dim=0) - g
