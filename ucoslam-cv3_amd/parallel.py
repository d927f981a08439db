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
* the fused frame stream (ShardedFrameStream, what `bench.py --gpus N` times beside the replicas): levels AND train tiles sharded,
  every rank holding ONLY its tile of the map, ONE all-gather per frame.  Matching frame t needs all of frame t's descriptors on
  every rank, i.e. it depends on the feature exchange; instead of a second collective per frame the stream is software-pipelined:
  the message of step t carries {this rank's level rows of frame t, its tile's accept lists for frame t-1, its slice of frame
  t-1's bag-of-words triplets} in one buffer, so results trail the input by one frame and every frame costs one collective.
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


# ------------------------------------------------------------------------------------------------ fused frame stream
def frame_message_layout(max_features: int, cand_cap: int, with_bow: bool = False):
    """Byte offsets of one rank's message (all multiples of 16).  Fields: header (int32: n level rows, n queries scanned),
    keypoints [max_features, 28 B], descriptors [max_features, 32 B], accept counts [max_features] int32, accept lists
    [max_features, cand_cap] uint64 (dist << 32 | global train index), optional bag-of-words triplets [max_features, 4] int32
    (word id, float weight bits, level node id, node valid) of this rank's slice of the previous frame's descriptors."""
    off, lay = 0, {}

    def take(name, nbytes):
        nonlocal off
        lay[name] = (off, nbytes)
        off += (nbytes + 15) & ~15

    take("header", 16)
    take("kps", max_features * 28)
    take("desc", max_features * 32)
    take("counts", max_features * 4)
    take("cand", max_features * cand_cap * 8)
    if with_bow:
        take("bow", max_features * 16)
    lay["total"] = off
    return lay


def pack_frame_message(buf, lay, kps, desc, n_rows: int, cand=None, counts=None, n_queries: int = 0, bow=None):
    """Fills `buf` (uint8 [lay.total], same device as the inputs).  kps [cap,7] float32, desc [cap,32] uint8 (first n_rows valid);
    cand [nq,cap] int64 / counts [nq] int32 for the PREVIOUS frame's queries (nq == n_queries), bow [slice,4] int32."""
    import torch

    o, n = lay["header"]
    if torch.is_tensor(n_rows):   # a device count (the extractor's own output): no host round trip, the whole capacity is copied
        hdr = buf[o:o + 16].view(torch.int32)
        hdr[0:1] = n_rows.reshape(1).to(torch.int32)
        hdr[1:2].fill_(n_queries)
        hdr[2:4].zero_()
        cap_rows = min(kps.shape[0], lay["kps"][1] // 28)
        o, n = lay["kps"]; buf[o:o + cap_rows * 28] = kps[:cap_rows].contiguous().view(torch.uint8).reshape(-1)
        o, n = lay["desc"]; buf[o:o + cap_rows * 32] = desc[:cap_rows].contiguous().reshape(-1)
    else:
        buf[o:o + 16] = torch.tensor([n_rows, n_queries, 0, 0], dtype=torch.int32).to(buf.device).view(torch.uint8)
        o, n = lay["kps"]; buf[o:o + n_rows * 28] = kps[:n_rows].contiguous().view(torch.uint8).reshape(-1)
        o, n = lay["desc"]; buf[o:o + n_rows * 32] = desc[:n_rows].contiguous().reshape(-1)
    if n_queries:
        o, n = lay["counts"]; buf[o:o + n_queries * 4] = counts[:n_queries].contiguous().view(torch.uint8).reshape(-1)
        o, n = lay["cand"]; buf[o:o + cand[:n_queries].numel() * 8] = cand[:n_queries].contiguous().view(torch.uint8).reshape(-1)
        if bow is not None and "bow" in lay:
            o, n = lay["bow"]; buf[o:o + bow.numel() * 4] = bow.contiguous().view(torch.uint8).reshape(-1)
    return buf


def unpack_frame_messages(all_buf, lay, world: int, max_features: int, cand_cap: int, bow_slices=None):
    """all_buf: uint8 [world, lay.total].  Returns dict(kps [n,7] float32, desc [n,32] uint8 — the level rows compacted in rank
    (= level) order —, cand_all [world, nq, cap] int64, counts_all [world, nq] int32, nq, bow [nq,4] int32 or None)."""
    import torch

    hdr = torch.stack([all_buf[r, lay["header"][0]:lay["header"][0] + 16].view(torch.int32) for r in range(world)]).cpu().numpy()
    rows, nq = hdr[:, 0].tolist(), int(hdr[0, 1])
    if max(rows) > max_features or min(rows) < 0 or not 0 <= nq <= max_features:   # a count above capacity would run into the next field
        raise ValueError(f"frame message header out of range: level rows {rows}, queries {nq}, capacity {max_features}")
    ko, do = lay["kps"][0], lay["desc"][0]
    kps = torch.cat([all_buf[r, ko:ko + rows[r] * 28].view(torch.float32).reshape(-1, 7) for r in range(world)], 0)
    desc = torch.cat([all_buf[r, do:do + rows[r] * 32].reshape(-1, 32) for r in range(world)], 0)
    out = {"kps": kps, "desc": desc, "nq": nq, "cand_all": None, "counts_all": None, "bow": None}
    if nq:
        co, ao = lay["counts"][0], lay["cand"][0]
        out["counts_all"] = torch.stack([all_buf[r, co:co + nq * 4].view(torch.int32) for r in range(world)])
        out["cand_all"] = torch.stack([all_buf[r, ao:ao + nq * cand_cap * 8].view(torch.int64).reshape(nq, cand_cap) for r in range(world)])
        if bow_slices is not None and "bow" in lay:
            bo = lay["bow"][0]
            out["bow"] = torch.cat([all_buf[r, bo:bo + (bow_slices[r + 1] - bow_slices[r]) * 16].view(torch.int32).reshape(-1, 4) for r in range(world)], 0)
    return out


def gather_messages(buf, group=None):
    """THE one collective of a frame: all-gather of the fixed-size message buffers -> uint8 [world, total]."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
    dist.all_gather([out[r] for r in range(world)], buf, group=group)
    return out


class ShardedFrameStream:
    """One frame stream over all ranks of `group` (BASELINE config 5): pyramid levels sharded, the map's train
    descriptors sharded into contiguous tiles (rank r holds rows [nt*r/world, nt*(r+1)/world) and nothing else), one fused
    all-gather per frame.  step(frame) returns the COMPLETE extraction of `frame` and the kNN rows (+ optional BoW triplets) of the
    PREVIOUS frame — identical on every rank and identical to the single-GPU results.

    tile_index: ucoslam_cv3_amd.knn.Index built over this rank's tile with set_row_offset(first global row).
    rank / world default to the process group's; tests pass them explicitly and drive local() / finish() themselves."""

    def __init__(self, extractor, params, tile_index, nn: int, max_features: int, cand_cap: int = 64, sorted: bool = False,
                 vocabulary=None, bow_level: int = 3, group=None, rank: int | None = None, world: int | None = None):
        self.ext, self.params, self.index, self.nn, self.sorted = extractor, params, tile_index, nn, sorted
        self.cap, self.maxf, self.voc, self.bow_level, self.group = cand_cap, max_features, vocabulary, bow_level, group
        if rank is None or world is None:
            import torch.distributed as dist

            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world = rank, world
        self.lay = frame_message_layout(max_features, cand_cap, vocabulary is not None)
        self._device_counts = True   # the level-row count travels as a device value (False: read back, rows copied exactly — tests)
        self.prev = None      # (kps, desc) of the previous frame, complete
        self._buf = None
        self._slices = None

    def local(self, frame):
        """This rank's share of step t: its levels of `frame`, its tile's accept lists and its BoW slice for frame t-1 -> message."""
        import torch

        H, W = frame.shape
        first, end = level_ranges(W, H, self.params.nOctaveLevels, self.params.scaleFactor, self.world)[self.rank]
        self.ext.setLevelRange(first, end)
        try:
            kps, desc, counts = self.ext.extract_batch(frame[None], self.params)
        finally:
            self.ext.setLevelRange(0, -1)
        if self._buf is None:
            self._buf = torch.zeros(self.lay["total"], dtype=torch.uint8, device=frame.device)
        cand = ccounts = bow = None
        nq, self._slices = 0, None
        if self.prev is not None:
            pq = self.prev[1]
            nq = pq.shape[0]
            cand, ccounts = self.index.scan_shard(pq, self.nn, self.cap)
            if self.voc is not None:
                self._slices = shard_bounds(nq, self.world)
                bow = self.voc.transform_triplets(pq[self._slices[self.rank]:self._slices[self.rank + 1]], self.bow_level)
        n_rows = counts[0] if self._device_counts else int(counts[0])   # (device count: no synchronisation before the collective)
        return pack_frame_message(self._buf, self.lay, kps[0], desc[0], n_rows, cand, ccounts, nq, bow)

    def finish(self, msgs):
        """msgs: uint8 [world, total] = every rank's message in rank order."""
        u = unpack_frame_messages(msgs, self.lay, self.world, self.maxf, self.cap, self._slices)
        result = {"kps": u["kps"], "desc": u["desc"], "prev_indices": None, "prev_distances": None, "prev_bow": u["bow"], "overflow": None}
        if u["nq"]:
            idx, dd, ovf = self.index.replay_tiles(self.prev[1], self.nn, u["cand_all"], u["counts_all"], sorted=self.sorted)
            result.update(prev_indices=idx, prev_distances=dd, overflow=ovf)
        self.prev = (u["kps"], u["desc"])
        return result

    def step(self, frame):
        return self.finish(gather_messages(self.local(frame), self.group))


# ------------------------------------------------------------------------------------------------ the same stream below Python
class _FStreamParams(__import__("ctypes").Structure):
    _fields_ = [(n, __import__("ctypes").c_int32) for n in ("rank", "world", "nn", "max_features", "cand_cap", "sorted", "bow_level")]


def _declare_fstream(L, sig):
    import ctypes as C

    from ._lib import I, VP

    sig("uh_fstream_create", I, VP, VP, VP, VP, C.POINTER(_FStreamParams), C.POINTER(VP))
    sig("uh_fstream_destroy", None, VP)
    sig("uh_fstream_message_bytes", C.c_size_t, VP)
    sig("uh_fstream_send_buffer", VP, VP)
    sig("uh_fstream_recv_buffer", VP, VP)
    sig("uh_fstream_comm_unique_id", I, VP)
    sig("uh_fstream_comm_init", I, VP, VP)
    sig("uh_fstream_set_comm", I, VP, VP)
    sig("uh_fstream_comm_ranks", I, VP)
    sig("uh_fstream_put_message", I, VP, I, VP)
    sig("uh_fstream_local_dev", I, VP, VP, I, I, C.c_size_t, I, I)
    sig("uh_fstream_exchange", I, VP)
    sig("uh_fstream_finish_dev", I, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP)


from . import _lib as _lib_mod  # noqa: E402

_lib_mod._EXTRA_DECLS.append(_declare_fstream)


class ShardedFrameStreamDev:
    """ShardedFrameStream on the C ABI (uh_fstream_*, csrc/fstream.hip): the producers write straight into the message, the replay reads
    the gathered lists in place, the collective is RCCL's all-gather on the context's stream through the library's own communicator
    (init_comm), counts stay on the device — a step is a handful of launches and never synchronises with the host.
    step(frame) returns device tensors: kps [F,7] / desc [F,32] / count [1] of `frame`, and prev_indices / prev_distances [F,nn] /
    prev_count [1] (+ bow_word / bow_weight / bow_node / bow_valid [F]) of the PREVIOUS frame; rows beyond a count are undefined."""

    def __init__(self, ctx, extractor, params, tile_index, nn: int, max_features: int, cand_cap: int = 64, sorted: bool = False,
                 vocabulary=None, bow_level: int = 3, rank: int = 0, world: int = 1, device=None):
        import ctypes as C

        import torch

        from ._lib import VP, check, lib

        self.ctx, self.ext, self.params, self.index, self.voc = ctx, extractor, params, tile_index, vocabulary
        self.rank, self.world, self.nn, self.F = rank, world, nn, max_features
        check(lib().uh_orb_set_params(extractor._h, C.byref(params)))
        p = _FStreamParams(rank, world, nn, max_features, cand_cap, int(sorted), bow_level)
        self._h = VP()
        check(lib().uh_fstream_create(ctx.handle, extractor._h, tile_index._h, vocabulary._h if vocabulary is not None else None, C.byref(p), C.byref(self._h)))
        self.message_bytes = int(lib().uh_fstream_message_bytes(self._h))
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        F = max_features
        self.out = dict(kps=torch.zeros((F, 7), dtype=torch.float32, device=dev), desc=torch.zeros((F, 32), dtype=torch.uint8, device=dev),
                        count=torch.zeros(1, dtype=torch.int32, device=dev), prev_indices=torch.zeros((F, nn), dtype=torch.int32, device=dev),
                        prev_distances=torch.zeros((F, nn), dtype=torch.int32, device=dev), prev_count=torch.zeros(1, dtype=torch.int32, device=dev),
                        overflow=torch.zeros(1, dtype=torch.int32, device=dev))
        if vocabulary is not None:
            self.out.update(bow_word=torch.zeros(F, dtype=torch.int32, device=dev), bow_weight=torch.zeros(F, dtype=torch.float32, device=dev),
                            bow_node=torch.zeros(F, dtype=torch.int32, device=dev), bow_valid=torch.zeros(F, dtype=torch.uint8, device=dev))
        self._ranges = {}

    def init_comm(self, group=None):
        """The library's own RCCL communicator over the ranks of `group`: rank 0 draws the unique id, torch.distributed carries the
        128 bytes (a host channel), every rank joins."""
        import ctypes as C

        import torch.distributed as dist

        from ._lib import check, lib

        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            check(lib().uh_fstream_comm_unique_id(buf))
            ident = [bytes(buf)]
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        buf = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        check(lib().uh_fstream_comm_init(self._h, buf))
        return self

    def comm_ranks(self) -> int:
        """Ranks of the RCCL communicator in use (ncclCommCount); 1 when the stream runs without one."""
        from ._lib import check, lib

        n = lib().uh_fstream_comm_ranks(self._h)
        if n < 0:
            check(n)
        return n

    def local(self, frame):
        from ._lib import check, dev_ptr, lib

        H, W = frame.shape
        key = (W, H)
        if key not in self._ranges:
            self._ranges[key] = level_ranges(W, H, self.params.nOctaveLevels, self.params.scaleFactor, self.world)[self.rank]
        first, end = self._ranges[key]
        check(lib().uh_fstream_local_dev(self._h, dev_ptr(frame), W, H, frame.stride(0), first, end))

    def exchange(self):
        from ._lib import check, lib

        check(lib().uh_fstream_exchange(self._h))

    def put_message(self, rank: int, other: "ShardedFrameStreamDev"):
        """(tests: several ranks played on one GPU) `other`'s message into this rank's receive buffer."""
        from ._lib import check, lib

        check(lib().uh_fstream_put_message(self._h, rank, lib().uh_fstream_send_buffer(other._h)))

    def finish(self):
        from ._lib import check, dev_ptr, lib

        o = self.out
        bow = [dev_ptr(o[k]) for k in ("bow_word", "bow_weight", "bow_node", "bow_valid")] if self.voc is not None else [None] * 4
        check(lib().uh_fstream_finish_dev(self._h, dev_ptr(o["kps"]), dev_ptr(o["desc"]), dev_ptr(o["count"]), dev_ptr(o["prev_indices"]),
                                          dev_ptr(o["prev_distances"]), dev_ptr(o["prev_count"]), *bow, dev_ptr(o["overflow"])))
        return o

    def step(self, frame):
        self.local(frame)
        self.exchange()
        return self.finish()

    def close(self):
        from ._lib import VP, lib

        if self._h:
            lib().uh_fstream_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
