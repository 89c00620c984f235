"""Multi-GPU plumbing: independent MPC problem instances sharded over ranks.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests). The only shared data are the read-only model weights: rank 0 broadcasts the blob once at
start-up. There is no per-solve or per-iteration collective — each instance owns its decision
variables — so scaling is weak: instances per GPU stay fixed as ranks are added.
"""
from __future__ import annotations

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed(force: bool = False) -> bool:
    """force: treat an initialised one-rank group as distributed (exercises the collectives on a single GPU)"""
    d = _dist()
    return d.is_available() and d.is_initialized() and (d.get_world_size() > 1 or force)


def broadcast_blob(blob, src: int = 0, device=None, force: bool = False) -> bytes:
    """Rank `src` supplies the model blob (bytes); every rank returns the same bytes."""
    import torch
    if not is_distributed(force):
        return bytes(blob)
    d = _dist()
    n = torch.tensor([len(blob) if d.get_rank() == src else 0], dtype=torch.int64, device=device)
    d.broadcast(n, src=src)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    if d.get_rank() == src:
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    d.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def shard_range(total: int, rank: int, world: int):
    """Contiguous block partition of `total` instances; the first total%world ranks get one extra."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def max_over_ranks(value: float, device=None, force: bool = False) -> float:
    import torch
    if not is_distributed(force):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return float(t.item())


def max_over_ranks_each(values, device=None, force: bool = False):
    """Element-wise maximum over ranks of equally long sequences (e.g. per-tick durations of barrier-aligned ticks)."""
    import torch
    if not is_distributed(force):
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return [float(v) for v in t.cpu().numpy()]


def gather_rows(local: np.ndarray, device=None) -> np.ndarray:
    """Concatenate per-rank result rows (e.g. uopt [b, H, m]) in rank order on every rank (reporting only)."""
    import torch
    if not is_distributed():
        return np.asarray(local)
    d = _dist()
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(d.get_world_size())]
    d.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=device))
    mx = int(max(int(c.item()) for c in counts))
    pad = np.zeros((mx,) + local.shape[1:], dtype=local.dtype)
    pad[:local.shape[0]] = local
    outs = [torch.zeros(pad.shape, dtype=torch.from_numpy(pad).dtype, device=device) for _ in range(d.get_world_size())]
    d.all_gather(outs, torch.from_numpy(pad).to(device) if device is not None else torch.from_numpy(pad))
    return np.concatenate([o.cpu().numpy()[:int(c.item())] for o, c in zip(outs, counts)], axis=0)


def sum_over_ranks(values, device=None, force: bool = False):
    """Element-wise integer sum over ranks (e.g. words of the timed launches that differ from the checker: must be 0 everywhere)."""
    import torch
    if not is_distributed(force):
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return [int(v) for v in t.cpu().numpy()]


def gather_int64(values, device=None, force: bool = False):
    """Every rank's equally long list of 64-bit integers, in rank order, on every rank (fingerprints of what a rank holds, per-rank counts)."""
    import torch
    if not is_distributed(force):
        return [[int(v) for v in values]]
    d = _dist()
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    outs = [torch.zeros_like(mine) for _ in range(d.get_world_size())]
    d.all_gather(outs, mine)
    return [[int(v) for v in o.cpu().numpy()] for o in outs]
