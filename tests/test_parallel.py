"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: shard split, candidate all-gather layout, and the
superset-replay argument, with the CPU oracle standing in for the per-shard GPU scan."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _local_accept_list(L, train, q, nn, t0, t1, cap):
    """What uh_knn_scan_shard_dev emits for rows [t0,t1): candidates the LOCAL heap accepted, in index order."""
    import oracle_lib

    nq = len(q)
    cand = np.zeros((nq, cap), np.int64)
    counts = np.zeros(nq, np.int32)
    for r in range(nq):
        # sequential replay of the accept rule on the shard: accepted iff fewer than nn earlier shard rows have dist <= d
        d = np.unpackbits(np.bitwise_xor(q[r][None, :], train[t0:t1]), axis=1).sum(1)
        kept = []
        import heapq
        worst = []   # max-heap of the nn best (negated)
        for i, di in enumerate(d.tolist()):
            if len(worst) < nn or di < -worst[0]:
                kept.append((di, t0 + i))
                if len(worst) == nn:
                    heapq.heapreplace(worst, -di)
                else:
                    heapq.heappush(worst, -di)
        counts[r] = len(kept)
        for j, (di, gi) in enumerate(kept[:cap]):
            cand[r, j] = (di << 32) | gi
    return cand, counts


def _replay(cand_all, counts_all, nn, sorted_):
    """The exact ResultSet replay over concatenated shard lists (what uh_knn_replay_dev does), in numpy/python."""
    world, nq, cap = cand_all.shape
    idx = np.full((nq, nn), -1, np.int32)
    dist_ = np.zeros((nq, nn), np.int32)
    for r in range(nq):
        hd, hi = [], []

        def swp(a, b):
            hd[a], hd[b] = hd[b], hd[a]
            hi[a], hi[b] = hi[b], hi[a]

        def up(i, n):
            while True:
                l, rr = 2 * i + 1, 2 * i + 2
                if l >= n:
                    return
                if rr >= n:
                    if hd[i] < hd[l]:
                        swp(i, l)
                    return
                if hd[rr] < hd[l]:
                    if hd[i] < hd[l]:
                        swp(i, l); i = l
                    else:
                        return
                else:
                    if hd[i] < hd[rr]:
                        swp(i, rr); i = rr
                    else:
                        return

        n = 0
        for s in range(world):
            assert counts_all[s, r] <= cap, "test sized so that no shard list overflows"
            for j in range(counts_all[s, r]):
                c = int(cand_all[s, r, j])
                d, gi = c >> 32, c & 0xFFFFFFFF
                if n >= nn:
                    if d < hd[0]:
                        swp(0, n - 1); n -= 1
                        if n > 1:
                            up(0, n)
                        hd.pop(); hi.pop()
                    else:
                        continue
                hd.append(d); hi.append(gi)
                k = n
                while k != 0:
                    p = (k - 1) // 2
                    if hd[p] < hd[k]:
                        swp(k, p); k = p
                    else:
                        break
                n += 1
        for j in range(n):
            idx[r, j], dist_[r, j] = hi[j], hd[j]
        if sorted_:
            for i in range(nn - 1):
                if idx[r, i] != -1:
                    for j in range(i + 1, nn):
                        if dist_[r, i] > dist_[r, j]:
                            dist_[r, i], dist_[r, j] = dist_[r, j], dist_[r, i]
                            idx[r, i], idx[r, j] = idx[r, j], idx[r, i]
    return idx, dist_


def _worker(rank, world, port, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    import synth
    from ucoslam_cv3_amd import parallel

    L = oracle_lib.load_oracle()
    train, q = synth.tie_stress_set(24, 400, seed=3) if rank >= 0 else None
    nn, cap = 4, 128
    b = parallel.shard_bounds(len(train), world)
    assert b[0] == 0 and b[-1] == len(train) and all(b[i] <= b[i + 1] for i in range(world))
    cand, counts = _local_accept_list(L, train, q, nn, b[rank], b[rank + 1], cap)
    cand_all, counts_all = parallel.gather_candidate_blocks(torch.from_numpy(cand), torch.from_numpy(counts))
    assert cand_all.shape == (world, len(q), cap) and counts_all.shape == (world, len(q))
    # every rank sees the same blocks, in shard order
    assert (cand_all[rank].numpy() == cand).all()
    for s in (0, 1):
        idx, dd = _replay(cand_all.numpy(), counts_all.numpy(), nn, s)
        ri, rd = oracle_lib.knn_search(L, train, q, nn, s)
        assert (idx == ri).all() and (dd == rd).all(), f"rank {rank}: sharded replay != unsharded reference (sorted={s})"
    # pyramid-level shards of ONE frame: every rank holds the rows of its contiguous level range (the oracle's rows with
    # those octaves stand in for the per-rank GPU extraction); one gather + compaction in rank order = the full extraction
    img = synth.frame(320, 240, seed=4)
    nf, nl, sf = 600, 6, 1.2
    rk, rd = oracle_lib.orb_extract(L, img, nf, nl, sf)
    ranges = parallel.level_ranges(320, 240, nl, sf, world)
    first, end = ranges[rank]
    mine = (rk["octave"] >= first) & (rk["octave"] < end)
    kp_blk = np.zeros((nf, 7), np.float32)
    ds_blk = np.zeros((nf, 32), np.uint8)
    n_mine = int(mine.sum())
    kp_blk[:n_mine] = rk[mine].view(np.float32).reshape(-1, 7)
    ds_blk[:n_mine] = rd[mine]
    gk, gd = parallel.gather_level_shards(torch.from_numpy(kp_blk), torch.from_numpy(ds_blk), torch.tensor(n_mine, dtype=torch.int32))
    assert gk.numpy().tobytes() == rk.tobytes() and (gd.numpy() == rd).all(), f"rank {rank}: gathered level shards != full extraction"
    fr = [list(parallel.frames_of_rank(10, r, world)) for r in range(world)]
    assert sum(fr, []) == list(range(10))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the timing reduction bench.py uses
    assert t.item() == world
    ok[rank] = 1
    dist.destroy_process_group()


def test_level_ranges_cover_the_pyramid_in_order():
    from ucoslam_cv3_amd import parallel

    for (w, h, nl, sf) in ((1241, 376, 8, 1.2), (640, 480, 8, 1.2), (320, 240, 3, 1.5), (97, 131, 1, 1.2)):
        for world in (1, 2, 3, 4, 8, 12):
            r = parallel.level_ranges(w, h, nl, sf, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == nl
            assert all(a <= b for a, b in r) and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    # one level per rank once there are as many ranks as levels; level 0 (a third of the pixels) is never paired at 4+
    assert parallel.level_ranges(1241, 376, 8, 1.2, 8) == [(i, i + 1) for i in range(8)]
    assert parallel.level_ranges(1241, 376, 8, 1.2, 4)[0] == (0, 1)


def test_gloo_world2_sharded_match_and_timing_reduction():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(1241, 376, 2000, 8, 1.2), (640, 480, 4000, 8, 1.2), (320, 240, 1000, 3, 1.5)], ids=lambda c: f"{c[0]}x{c[1]}_l{c[3]}")
def test_level_shards_concatenate_to_the_full_extraction(hip_ctx, cfg):
    """uh_orb_set_level_range: what ranks 0..world-1 of a 2/3/8/11-GPU extraction would each produce (run one after the other
    on the one GPU here), concatenated in rank order, equals the single-GPU extraction (itself bit-exact vs the oracle)."""
    import oracle_lib
    import synth
    from ucoslam_cv3_amd import parallel
    from ucoslam_cv3_amd.orb import KEYPOINT_DTYPE, FeatParams, ORBextractor

    w, h, nf, nl, sf = cfg
    img = synth.frame(w, h, seed=21)
    rk, rd = oracle_lib.orb_extract(oracle_lib.load_oracle(), img, nf, nl, sf)
    frame = torch.from_numpy(img).cuda()
    ext = ORBextractor.create(hip_ctx)
    fp = FeatParams(nf, nl, sf)
    for world in (1, 2, 3, 8, 11):
        ks, ds = [], []
        for first, end in parallel.level_ranges(w, h, nl, sf, world):
            ext.setLevelRange(first, end)
            kps, desc, counts = ext.extract_batch(frame[None], fp)
            torch.cuda.synchronize()
            n = int(counts[0])
            got = kps[0, :n].cpu().numpy()
            assert first < end or n == 0
            if n:
                oct_ = got.copy().view(KEYPOINT_DTYPE)["octave"]
                assert oct_.min() >= first and oct_.max() < end
            ks.append(got)
            ds.append(desc[0, :n].cpu().numpy())
        ext.setLevelRange(0, -1)
        k_all, d_all = np.concatenate(ks), np.concatenate(ds)
        assert k_all.tobytes() == rk.tobytes(), f"world {world}: keypoints of the level shards != full extraction"
        np.testing.assert_array_equal(d_all, rd, err_msg=f"world {world}")


@pytest.mark.gpu
def test_sharded_extract_over_rccl_single_rank(hip_ctx):
    """parallel.sharded_extract through the RCCL backend with world_size 1 (one device on the GPU box)."""
    import oracle_lib
    import synth
    from ucoslam_cv3_amd import parallel
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        img = synth.frame(640, 480, seed=2)
        ext = ORBextractor.create(hip_ctx)
        gk, gd = parallel.sharded_extract(ext, torch.from_numpy(img).cuda(), FeatParams(2000, 8, 1.2))
        torch.cuda.synchronize()
        rk, rd = oracle_lib.orb_extract(oracle_lib.load_oracle(), img, 2000, 8, 1.2)
        assert gk.cpu().numpy().tobytes() == rk.tobytes() and (gd.cpu().numpy() == rd).all()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_search_over_rccl_single_rank(hip_ctx):
    """The RCCL ("nccl") code path with world_size 1 (the GPU box has one device): scan -> all_gather -> replay."""
    import oracle_lib
    import synth
    from ucoslam_cv3_amd import parallel
    from ucoslam_cv3_amd.knn import Index

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        train, q = synth.match_set(150, 2000, seed=8)
        index = Index(hip_ctx).build(torch.from_numpy(train).cuda())
        idx, dd = parallel.sharded_search(index, torch.from_numpy(q).cuda(), 10, sorted=False, cap=128)
        torch.cuda.synchronize()
        ri, rd = oracle_lib.knn_search(oracle_lib.load_oracle(), train, q, 10, 0)
        assert (idx.cpu().numpy() == ri).all() and (dd.cpu().numpy() == rd).all()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ fused frame stream (one all-gather per frame)
def _fused_worker(rank, world, port, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ucoslam_cv3_amd import parallel

    maxf, cap = 300, 16
    lay = parallel.frame_message_layout(maxf, cap, with_bow=True)
    assert all(o % 16 == 0 for k, (o, n) in ((k, v) for k, v in lay.items() if k != "total"))

    def rank_data(r):   # what rank r contributes: deterministic, so that every rank can check everybody's part
        g = np.random.default_rng(100 + r)
        rows = 40 + 17 * r
        kps = g.standard_normal((maxf, 7)).astype(np.float32)
        desc = g.integers(0, 256, (maxf, 32), dtype=np.uint8)
        nq = 123
        cand = g.integers(0, 2 ** 62, (nq, cap), dtype=np.int64)
        counts = g.integers(0, cap + 1, nq).astype(np.int32)
        sl = parallel.shard_bounds(nq, world)
        bow = g.integers(-2 ** 31, 2 ** 31 - 1, (sl[r + 1] - sl[r], 4)).astype(np.int32)
        return rows, kps, desc, nq, cand, counts, bow

    rows, kps, desc, nq, cand, counts, bow = rank_data(rank)
    buf = torch.zeros(lay["total"], dtype=torch.uint8)
    parallel.pack_frame_message(buf, lay, torch.from_numpy(kps), torch.from_numpy(desc), rows, torch.from_numpy(cand), torch.from_numpy(counts), nq,
                                torch.from_numpy(bow))
    msgs = parallel.gather_messages(buf)            # ONE collective
    assert msgs.shape == (world, lay["total"])
    u = parallel.unpack_frame_messages(msgs, lay, world, maxf, cap, parallel.shard_bounds(nq, world))
    exp = [rank_data(r) for r in range(world)]
    np.testing.assert_array_equal(u["kps"].numpy(), np.concatenate([e[1][: e[0]] for e in exp]))
    np.testing.assert_array_equal(u["desc"].numpy(), np.concatenate([e[2][: e[0]] for e in exp]))
    assert u["nq"] == nq
    np.testing.assert_array_equal(u["cand_all"].numpy(), np.stack([e[4] for e in exp]))
    np.testing.assert_array_equal(u["counts_all"].numpy(), np.stack([e[5] for e in exp]))
    np.testing.assert_array_equal(u["bow"].numpy(), np.concatenate([e[6] for e in exp]))
    ok[rank] = 1
    dist.destroy_process_group()


def test_gloo_world2_fused_frame_message_one_collective():
    world = 2
    port = 31500 + (os.getpid() % 2000)
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_fused_worker, args=(world, port, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_frame_stream_equals_single_gpu(hip_ctx, world):
    """ShardedFrameStream with `world` ranks simulated one after the other on this GPU (each with ONLY its tile of the map and its
    level range; the all-gather replaced by stacking the messages): complete features of frame t, kNN rows and bag-of-words
    triplets of frame t-1 — all identical to the single-GPU extraction / search / descent."""
    import synth
    from ucoslam_cv3_amd import parallel
    from ucoslam_cv3_amd.bow import Vocabulary, write_vocabulary_stream
    from ucoslam_cv3_amd.knn import Index
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    W, H, nf, nn = 640, 480, 1000, 10
    fp = FeatParams(nf, 8, 1.2)
    train, _ = synth.match_set(1, 3001, seed=5)
    d_train = torch.from_numpy(train).cuda()
    full = Index(hip_ctx).build(d_train)
    params, blob, _ = synth.vocabulary(k=10, depth=4, seed=2)
    voc = Vocabulary(hip_ctx).fromStream(write_vocabulary_stream(params, blob))
    ext = ORBextractor.create(hip_ctx)
    b = parallel.shard_bounds(len(train), world)
    streams = []
    for r in range(world):
        tile = Index(hip_ctx).build(d_train[b[r]:b[r + 1]].clone()).set_row_offset(b[r])
        streams.append(parallel.ShardedFrameStream(ext, fp, tile, nn, nf, cand_cap=96, vocabulary=voc, bow_level=3, rank=r, world=world))
    prev_desc = None
    for t in range(3):
        frame = torch.from_numpy(synth.frame(W, H, seed=40 + t, shift=(3 * t, t))).cuda()
        msgs = torch.stack([s.local(frame).clone() for s in streams])
        res = [s.finish(msgs) for s in streams]
        torch.cuda.synchronize()
        kps, desc, counts = ext.extract_batch(frame[None], fp)
        n = int(counts[0])
        for r in res:
            assert r["kps"].cpu().numpy().tobytes() == kps[0, :n].cpu().numpy().tobytes()
            assert (r["desc"] == desc[0, :n]).all()
        if prev_desc is not None:
            ri, rd = full.search(prev_desc, nn, sorted=False)
            trip = voc.transform_triplets(prev_desc, 3)
            for r in res:
                assert int(r["overflow"]) == 0
                assert (r["prev_indices"] == ri).all() and (r["prev_distances"] == rd).all()
                assert (r["prev_bow"] == trip).all()
            f1, f2 = Vocabulary.maps_from_triplets(res[0]["prev_bow"].cpu().numpy())
            g1, g2 = voc.transform(prev_desc.cpu().numpy(), 3)
            assert f1 == g1 and f2 == g2
        else:
            assert all(r["prev_indices"] is None for r in res)
        prev_desc = desc[0, :n].clone()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(640, 480, 1000, 3001, 2), (640, 480, 1000, 3001, 3), (1241, 376, 2000, 10000, 8), (1241, 376, 2000, 10000, 1)],
                         ids=lambda c: f"{c[0]}x{c[1]}_{c[2]}f_{c[3]}rows_world{c[4]}")
def test_sharded_frame_stream_below_python_equals_single_gpu(hip_ctx, cfg):
    """The C-level stream (uh_fstream_*, csrc/fstream.hip: producers write into the message, the replay reads the gathered lists in
    place, counts never leave the device) with `world` ranks played one after the other on this GPU — each with ONLY its tile of the
    map, its own extractor and its level range; the all-gather replaced by uh_fstream_put_message — at BASELINE config 5's size
    (1241 x 376, 2000 features, 10 000 map rows, 8 ranks) and smaller: features of frame t, kNN rows and fbow triples of frame t-1
    identical to the single-GPU extraction / search / descent on every rank."""
    import synth
    from ucoslam_cv3_amd import parallel
    from ucoslam_cv3_amd.bow import Vocabulary, write_vocabulary_stream
    from ucoslam_cv3_amd.knn import Index
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    W, H, nf, nt, world = cfg
    nn = 10
    fp = FeatParams(nf, 8, 1.2)
    train, _ = synth.match_set(1, nt, seed=5)
    d_train = torch.from_numpy(train).cuda()
    full = Index(hip_ctx).build(d_train)
    params, blob, _ = synth.vocabulary(k=10, depth=4, seed=2)
    voc = Vocabulary(hip_ctx).fromStream(write_vocabulary_stream(params, blob))
    ref_ext = ORBextractor.create(hip_ctx)
    b = parallel.shard_bounds(len(train), world)
    streams = []
    for r in range(world):
        tile = Index(hip_ctx).build(d_train[b[r]:b[r + 1]].clone()).set_row_offset(b[r])
        streams.append(parallel.ShardedFrameStreamDev(hip_ctx, ORBextractor.create(hip_ctx), fp, tile, nn, nf, cand_cap=96, vocabulary=voc, bow_level=3,
                                                      rank=r, world=world))
    prev_desc = None
    for t in range(4):
        frame = torch.from_numpy(synth.frame(W, H, seed=40 + t, shift=(3 * t, t))).cuda()
        for s in streams:
            s.local(frame)
        if world == 1:
            streams[0].exchange()                      # (world 1: the library's own path)
        else:
            for dst in streams:
                for r, src in enumerate(streams):
                    dst.put_message(r, src)
        res = [s.finish() for s in streams]
        torch.cuda.synchronize()
        kps, desc, counts = ref_ext.extract_batch(frame[None], fp)
        n = int(counts[0])
        for o in res:
            assert int(o["count"]) == n and int(o["overflow"]) == 0
            assert o["kps"][:n].cpu().numpy().tobytes() == kps[0, :n].cpu().numpy().tobytes()
            assert (o["desc"][:n] == desc[0, :n]).all()
        if prev_desc is not None:
            m = prev_desc.shape[0]
            ri, rd = full.search(prev_desc, nn, sorted=False)
            trip = voc.transform_triplets(prev_desc, 3)
            for o in res:
                assert int(o["prev_count"]) == m
                assert (o["prev_indices"][:m] == ri).all() and (o["prev_distances"][:m] == rd).all()
                got = torch.stack([o["bow_word"][:m], o["bow_weight"][:m].view(torch.int32), o["bow_node"][:m], o["bow_valid"][:m].to(torch.int32)], 1)
                v = trip[:, 3] != 0
                assert (got[:, 0] == trip[:, 0]).all() and (got[:, 1] == trip[:, 1]).all() and (got[:, 3] == trip[:, 3]).all()
                assert (got[v, 2] == trip[v, 2]).all()
        else:
            assert all(int(o["prev_count"]) == 0 for o in res)
        prev_desc = desc[0, :n].clone()


@pytest.mark.gpu
def test_fstream_exchange_through_rccl_with_one_rank(hip_ctx):
    """The library's own RCCL path (librccl resolved with dlopen, ncclGetUniqueId -> ncclCommInitRank -> ncclAllGather on the context's
    stream) with the one rank a 1-GPU box has: the results must equal the world-1 copy path's.  (More ranks need more GPUs: the
    driver's multi-GPU bench is the first run of that.)"""
    import ctypes as C

    import synth
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd import parallel
    from ucoslam_cv3_amd.knn import Index
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    fp = FeatParams(1000, 8, 1.2)
    train, _ = synth.match_set(1, 2000, seed=9)
    d_train = torch.from_numpy(train).cuda()
    outs = []
    for use_rccl in (False, True):
        s = parallel.ShardedFrameStreamDev(hip_ctx, ORBextractor.create(hip_ctx), fp, Index(hip_ctx).build(d_train), 10, 1000, cand_cap=96)
        if use_rccl:
            ident = (C.c_uint8 * 128)()
            u._lib.check(u.lib().uh_fstream_comm_unique_id(ident))
            u._lib.check(u.lib().uh_fstream_comm_init(s._h, ident))
        res = []
        for t in range(3):
            o = s.step(torch.from_numpy(synth.frame(640, 480, seed=70 + t, shift=(2 * t, t))).cuda())
            torch.cuda.synchronize()
            n, m = int(o["count"]), int(o["prev_count"])
            res.append((n, m, o["kps"][:n].cpu().numpy().tobytes(), o["desc"][:n].cpu().numpy().tobytes(), o["prev_indices"][:m].cpu().numpy().tobytes(),
                        o["prev_distances"][:m].cpu().numpy().tobytes(), int(o["overflow"])))
        outs.append(res)
    assert outs[0] == outs[1] and outs[0][2][0] > 300 and outs[0][2][1] > 300
