"""Sharding of a configuration batch over ranks (one process per GPU).

Configurations are independent, so the batch splits into contiguous ranges with no data-path
collective; an edge stays whole inside one shard because its first-bad index is
a reduction over its own samples.  The only exchange is the result gather: flag bytes (and
per-edge first-bad indices) over RCCL, plus a sum of counters.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank `rank`: sizes differ by at most one, ranks in order."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_edges(edge_off: Sequence[int], world: int, rank: int) -> Tuple[int, int, int, int]:
    """Edge-aligned shard: (edge_lo, edge_hi, sample_lo, sample_hi), balancing SAMPLES not edges."""
    edge_off = np.asarray(edge_off, np.int64)
    n_edges, total = len(edge_off) - 1, int(edge_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        e = int(np.searchsorted(edge_off, target, side="left"))
        cuts.append(min(max(e, cuts[-1]), n_edges))
    cuts.append(n_edges)
    e_lo, e_hi = cuts[rank], cuts[rank + 1]
    return e_lo, e_hi, int(edge_off[e_lo]), int(edge_off[e_hi])


def all_gather_ragged(dist, local, sizes: Sequence[int]):
    """Gathers per-rank 1-D tensors of the given sizes into one tensor on every rank."""
    import torch
    m = max(sizes)
    pad = torch.zeros(m, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    out = torch.empty(m * len(sizes), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(len(sizes))])
