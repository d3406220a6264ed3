"""Multi-GPU layer of the engine: sentences are independent (the reference evaluates them in a
sequential loop, bert.cpp:750), so the path shards embarrassingly — weights replicated on every
GPU, one process per GPU, each rank evaluates a contiguous range of sentences balanced by token
count, and the ONLY exchange step is one all-gather of the final [n_sentences, n_embd] embeddings
(RCCL over xGMI when the process group backend is "nccl"; "gloo" on CPU in the tests).

No collective runs inside the forward pass.  With 8 ranks in one node a direct all-gather puts each
peer's shard on its own xGMI link (7 links x ~153 GB/s per GPU): 125,000 x 768 f32 = 384 MB per rank
(BASELINE config 5) is ~2.5 ms per peer transfer against seconds of compute.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def shard_bounds(lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) sentence ranges per rank with near-equal TOKEN counts.
    Every sentence belongs to exactly one rank; ranks may be empty when world > n_sentences."""
    n = len(lengths)
    if world <= 0:
        raise ValueError("world must be positive")
    cum = np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])
    total = int(cum[-1])
    bounds, start = [], 0
    for r in range(world):
        if r == world - 1:
            end = n
        else:
            target = total * (r + 1) / world
            end = int(np.searchsorted(cum, target, side="left"))
            # cum[end] >= target; pick the closer of end-1 / end, never move backwards
            if end > 0 and abs(cum[end - 1] - target) <= abs(cum[min(end, n)] - target):
                end -= 1
            end = max(start, min(end, n))
        bounds.append((start, end))
        start = end
    return bounds


def gather_embeddings(local, counts: Sequence[int], group=None, out=None, async_op=False):
    """All-gather variable-sized shards of embeddings.

    local: torch tensor [counts[rank], H] on the rank's device; returns [sum(counts), H] in global
    sentence order on every rank.  One collective: shards are padded to the largest count so a
    single all_gather_into_tensor moves everything.  `out` (optional, equal shards only) is a
    preallocated [sum(counts), H] result buffer.
    async_op (equal shards only): returns (out, work) without waiting — the exchange of one batch runs under the forward
    pass of the next one; the caller waits on `work` before it reuses `local` or reads `out`."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    assert len(counts) == world
    H = local.shape[1]
    if min(counts) == max(counts) and local.shape[0] == counts[0] and counts[0] > 0:
        # equal shards (fixed-length batches, the benchmark case): no padding, no copies
        if out is None:
            out = torch.empty((world * counts[0], H), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
        return (out, work) if async_op else out
    assert not async_op, "gather_embeddings(async_op=True) needs equal shards"
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, H), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx, H), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx: r * mx + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def encode_sharded(eval_fn: Callable[[List[np.ndarray]], np.ndarray], sentences: Sequence[np.ndarray], device="cpu",
                   group=None):
    """Evaluate `sentences` (token-id arrays) across the process group and return all embeddings
    [n_sentences, H] on every rank.  eval_fn maps this rank's list of sentences to [n_local, H] f32
    (on a GPU box: BertModel.eval_batch / eval_packed of the rank's own context)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds([len(s) for s in sentences], world)
    s, e = bounds[rank]
    counts = [b - a for a, b in bounds]
    if e > s:
        local = torch.from_numpy(np.ascontiguousarray(eval_fn(list(sentences[s:e])), dtype=np.float32))
        H = local.shape[1]
    else:
        local, H = None, 0
    # ranks with an empty shard learn H from the others
    h = torch.tensor([H], dtype=torch.int64, device=device)
    dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
    H = int(h.item())
    if local is None:
        local = torch.zeros((0, H), dtype=torch.float32)
    return gather_embeddings(local.to(device), counts, group)
