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
    fr = [list(parallel.frames_of_rank(10, r, world)) for r in range(world)]
    assert sum(fr, []) == list(range(10))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the timing reduction bench.py uses
    assert t.item() == world
    ok[rank] = 1
    dist.destroy_process_group()


def test_gloo_world2_sharded_match_and_timing_reduction():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


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
