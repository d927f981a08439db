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
* sharded extraction of ONE frame: the pyramid levels are split into contiguous ranges, one per rank (the budgets, thresholds
  and cell grids of a level do not depend on the other levels' content, ORBextractor.cpp:501-513, and the outputs are
  concatenated in level order, :1286-1300), every rank builds the resize chain up to its last level redundantly (level l
  reads level l-1, :1379) and extracts its levels, ONE all-gather of the fixed-capacity (keypoint, descriptor) row blocks
  + counts moves <= maxFeatures * 60 B per rank, and every rank compacts the blocks in rank (= level) order — rows identical
  to the single-GPU extraction.  Level 0 alone holds 32 % of the pyramid's pixels, so this form cannot scale beyond ~3x
  and costs a collective per frame; it exists for latency (one frame, many GPUs), not for throughput.
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


def level_ranges(w: int, h: int, nlevels: int, scale_factor: float, world: int):
    """Contiguous level ranges [(first, end)] * world that minimise the largest per-rank cost (exact, by search over the
    splits; nlevels <= 16).  Cost of a range = pixels of the chain this rank must BUILD (levels 0..end-1: blur + resize) plus
    twice the pixels of the levels it extracts (FAST strength map + cell maxima dominate).  Ranks beyond the number of
    levels get empty ranges (first == end)."""
    import functools

    import numpy as np

    inv = np.ones(nlevels, np.float32)
    sc = np.float32(1.0)
    for i in range(1, nlevels):
        sc = np.float32(sc * np.float32(scale_factor))
        inv[i] = np.float32(1.0) / sc
    px = [int(np.rint(np.float32(w) * inv[i])) * int(np.rint(np.float32(h) * inv[i])) for i in range(nlevels)]
    pre = [0]
    for v in px:
        pre.append(pre[-1] + v)

    def cost(a, b):
        return 0 if a == b else pre[b] + 2 * (pre[b] - pre[a])

    @functools.lru_cache(maxsize=None)
    def best(first, parts):
        # -> (max cost, tuple of ends) covering levels [first, nlevels) with `parts` contiguous (possibly empty) ranges
        if parts == 1:
            return cost(first, nlevels), (nlevels,)
        out = None
        for end in range(first, nlevels + 1):
            tail = best(end, parts - 1)
            c = max(cost(first, end), tail[0])
            if out is None or c < out[0]:
                out = (c, (end,) + tail[1])
        return out

    ends = best(0, world)[1]
    firsts = (0,) + ends[:-1]
    return list(zip(firsts, ends))


def gather_level_shards(kps, desc, count, group=None):
    """ONE frame's per-rank extraction results -> the full extraction on every rank.

    kps [cap, 7] float32 (28-byte cv::KeyPoint rows), desc [cap, 32] uint8, count: 0-d / [1] int32 tensor with the rows this
    rank filled.  all-gather of the fixed-capacity blocks + counts, then compaction in rank order (= level order, because the
    level ranges are contiguous and ascending).  Returns (kps [n, 7], desc [n, 32])."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cap = kps.shape[0]
    k_all = torch.empty((world, cap, 7), dtype=kps.dtype, device=kps.device)
    d_all = torch.empty((world, cap, 32), dtype=desc.dtype, device=desc.device)
    c_all = torch.empty((world, 1), dtype=torch.int32, device=kps.device)
    dist.all_gather([c_all[r] for r in range(world)], count.reshape(1).to(torch.int32).contiguous(), group=group)
    dist.all_gather([k_all[r] for r in range(world)], kps.contiguous(), group=group)
    dist.all_gather([d_all[r] for r in range(world)], desc.contiguous(), group=group)
    n = c_all.reshape(-1).tolist()
    return (torch.cat([k_all[r, : n[r]] for r in range(world)], 0), torch.cat([d_all[r, : n[r]] for r in range(world)], 0))


def sharded_extract(extractor, frame, params, group=None):
    """Keypoints/descriptors of ONE frame (torch uint8 CUDA tensor [H, W]), pyramid levels sharded over the ranks of `group`.
    Identical on every rank and identical to extractor.extract_batch(frame[None], params) on one GPU."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    H, W = frame.shape
    first, end = level_ranges(W, H, params.nOctaveLevels, params.scaleFactor, world)[rank]
    extractor.setLevelRange(first, end)
    try:
        kps, desc, counts = extractor.extract_batch(frame[None], params)
    finally:
        extractor.setLevelRange(0, -1)
    return gather_level_shards(kps[0], desc[0], counts[0], group)


def frames_of_rank(n_frames: int, rank: int, world: int):
    """Contiguous block of a frame stream handled by `rank` (frame-parallel extraction)."""
    b = shard_bounds(n_frames, world)
    return range(b[rank], b[rank + 1])
