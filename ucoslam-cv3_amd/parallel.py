"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Two ways the hot path spreads over the GPUs of a node:

* frame streams (what bench.py times): extraction + matching + BA of different frames/sequences are independent, so every rank
  runs the whole per-frame path on its own frames — no data-path collective, weak scaling.
* sharded matching of ONE frame against a large map: the train descriptors are split into contiguous row shards (global
  indices kept, so the reference's lowest-index tie rule survives), every rank scans its shard for all queries and emits the
  candidates its local heap accepted, ONE all-gather moves the fixed-size candidate blocks, and every rank replays the
  concatenation through the exact ResultSet semantics (uh_knn_replay_dev) — bit-identical to the unsharded search.
  Message size: nq * cap * 8 B + nq * 4 B per rank (cap 64, nq 2000 -> ~1 MB): latency-bound on xGMI, hence one fused
  all-gather per search rather than a ring of small messages.
"""
from __future__ import annotations


def shard_bounds(nt: int, nshards: int):
    """Shard s = rows [nt*s//n, nt*(s+1)//n) — the split uh_knn_replay_dev assumes (csrc/knn.hip)."""
    return [nt * s // nshards for s in range(nshards + 1)]


def gather_candidate_blocks(cand, counts, group=None):
    """all-gather of the per-rank candidate lists; returns ([world, nq, cap] int64, [world, nq] int32) in rank (= shard) order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cand_all = torch.empty((world,) + tuple(cand.shape), dtype=cand.dtype, device=cand.device)
    counts_all = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device)
    # list form: accepted by both the RCCL ("nccl") and the gloo backend; the views alias cand_all / counts_all
    dist.all_gather([cand_all[r] for r in range(world)], cand.contiguous(), group=group)
    dist.all_gather([counts_all[r] for r in range(world)], counts.contiguous(), group=group)
    return cand_all, counts_all


def sharded_search(index, queries, nn: int, sorted: bool = False, cap: int = 64, max_dist: int = -1, group=None):
    """Exact kNN of `queries` against the FULL train set held by `index` (every rank holds all rows; each scans only its shard).

    index: ucoslam_cv3_amd.knn.Index built over the same rows on every rank.  Returns (indices, distances) identical on all
    ranks and identical to index.search(queries, nn, sorted)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b = shard_bounds(index.size(), world)
    index.set_shard(b[rank], b[rank + 1])
    cand, counts = index.scan_shard(queries, nn, cap, max_dist)
    index.set_shard(0, index.size())
    cand_all, counts_all = gather_candidate_blocks(cand, counts, group)
    return index.replay(queries, nn, cand_all, counts_all, sorted=sorted, max_dist=max_dist)


def frames_of_rank(n_frames: int, rank: int, world: int):
    """Contiguous block of a frame stream handled by `rank` (frame-parallel extraction)."""
    b = shard_bounds(n_frames, world)
    return range(b[rank], b[rank + 1])
