"""Hamming kNN: oracle vs the real xflann (golden + live _ref), and HIP vs oracle (gpu)."""
import os

import numpy as np
import pytest

import oracle_lib
import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "knn_golden.npz")
CASES = ["rand", "ties", "desc", "tiny"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nn", [1, 2, 10])
@pytest.mark.parametrize("s", [0, 1])
def test_oracle_matches_reference_golden(oracle, case, nn, s):
    g = np.load(GOLD)
    idx, dist = oracle_lib.knn_search(oracle, g[f"{case}_train"], g[f"{case}_q"], nn, s)
    np.testing.assert_array_equal(idx, g[f"{case}_nn{nn}_s{s}_idx"])
    np.testing.assert_array_equal(dist, g[f"{case}_nn{nn}_s{s}_dist"])


def test_oracle_matches_live_reference_build(oracle):
    ref = oracle_lib.load_ref("xflann")
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent on this box)")
    P = oracle_lib.P
    for seed in range(3):
        train, q = synth.match_set(150, 2000, seed=100 + seed)
        for nn, s in [(2, 1), (10, 0), (7, 1)]:
            i2 = np.empty((len(q), nn), np.int32)
            d2 = np.empty_like(i2)
            assert ref.xflann_ref_linear_search(P(train), len(train), P(q), len(q), nn, s, 1, P(i2), P(d2)) == 0
            i1, d1 = oracle_lib.knn_search(oracle, train, q, nn, s)
            np.testing.assert_array_equal(i1, i2)
            np.testing.assert_array_equal(d1, d2)


def test_oracle_shard_superset_property(oracle):
    """Top-1 over the whole set == best of per-shard top-1 (lowest index wins ties): host logic of the sharded path."""
    train, q = synth.tie_stress_set(40, 600, seed=5)
    full_i, full_d = oracle_lib.knn_search(oracle, train, q, 1)
    parts = [oracle_lib.knn_search(oracle, train, q, 1, t_begin=a, t_end=b) for a, b in [(0, 200), (200, 450), (450, 600)]]
    best_d = np.minimum.reduce([p[1][:, 0] for p in parts])
    np.testing.assert_array_equal(best_d, full_d[:, 0])
    for r in range(len(q)):
        cand = [p[0][r, 0] for p in parts if p[1][r, 0] == best_d[r]]
        assert full_i[r, 0] == min(cand)


# ------------------------------------------------------------------ GPU parity (through the C ABI)
def _gpu_cases():
    return {
        "rand": synth.match_set(300, 3000, seed=21),
        "ties": synth.tie_stress_set(200, 1500, seed=22),
        "desc": synth.descending_set(8, 700, seed=23),
        "tiny": synth.match_set(7, 3, seed=24),
        "one": synth.match_set(65, 1, seed=25),
        "ragged": synth.match_set(129, 257, seed=26),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rand", "ties", "desc", "tiny", "one", "ragged"])
def test_hip_knn_bit_exact_host_api(hip_ctx, oracle, case):
    from ucoslam_cv3_amd.knn import Index

    train, q = _gpu_cases()[case]
    index = Index(hip_ctx).build(train)
    assert index.size() == len(train)
    for nn in (1, 2, 10, 33):
        for s in (0, 1):
            idx, dist = index.search(q, nn, sorted=bool(s))
            ri, rd = oracle_lib.knn_search(oracle, train, q, nn, s)
            np.testing.assert_array_equal(idx, ri, err_msg=f"{case} nn={nn} sorted={s}")
            np.testing.assert_array_equal(dist, rd)


@pytest.mark.gpu
def test_hip_knn_radius_bound_and_strided_rows(hip_ctx, oracle):
    from ucoslam_cv3_amd.knn import Index

    train, q = synth.match_set(100, 900, seed=31)
    big_t = np.zeros((len(train), 48), np.uint8)
    big_t[:, :32] = train
    big_q = np.zeros((len(q), 40), np.uint8)
    big_q[:, :32] = q
    index = Index(hip_ctx).build(big_t[:, :32])
    for md in (0, 60, 100, 120):
        idx, dist = index.search(big_q[:, :32], 5, sorted=True, max_dist=md)
        ri, rd = oracle_lib.knn_search(oracle, train, q, 5, 1, max_dist=md)
        np.testing.assert_array_equal(idx, ri)
        np.testing.assert_array_equal(dist, rd)


@pytest.mark.gpu
@pytest.mark.parametrize("qpw", [2, 4])
def test_hip_knn_queries_per_wave_identical_rows(hip_ctx, oracle, qpw):
    """uh_knn_set_queries_per_wave: 2 / 4 queries share the train rows a wave loads; rows (heap order included) stay the reference's,
    for query counts that are not multiples of the group, tie-heavy sets, nn = 1 (depth-1 heap), 2, 10, 15 and a radius bound."""
    from ucoslam_cv3_amd.knn import Index

    for case, (train, q) in _gpu_cases().items():
        index = Index(hip_ctx).build(train).set_queries_per_wave(qpw)
        for nn in (1, 2, 10, 15, 33):            # 33 > 15: served by the one-query kernel
            for s in (0, 1):
                idx, dist = index.search(q, nn, sorted=bool(s))
                ri, rd = oracle_lib.knn_search(oracle, train, q, nn, s)
                np.testing.assert_array_equal(idx, ri, err_msg=f"{case} nn={nn} sorted={s} qpw={qpw}")
                np.testing.assert_array_equal(dist, rd)
    train, q = synth.match_set(1003, 9999, seed=77)
    index = Index(hip_ctx).build(train).set_queries_per_wave(qpw)
    for md in (-1, 70):
        idx, dist = index.search(q, 10, sorted=False, max_dist=md)
        ri, rd = oracle_lib.knn_search(oracle, train, q, 10, 0, max_dist=md)
        np.testing.assert_array_equal(idx, ri)
        np.testing.assert_array_equal(dist, rd)
    import ucoslam_cv3_amd as u
    with pytest.raises(u.UcoslamHipError):
        index.set_queries_per_wave(3)


@pytest.mark.gpu
@pytest.mark.parametrize("form,accept_qpw", [("twophase", "1"), ("twophase", "2"), ("stream", "2")])
def test_hip_knn_two_phase_form_every_k(hip_ctx, oracle, form, accept_qpw, monkeypatch):
    """The accept-list scan + lane-per-query replay (default from 6000 queries on, forced here for every size), as two launches and as
    one launch whose replay waves consume the lists while the scan appends to them: every k the register heap is instantiated for
    (1..16), sorted and unsorted, ties, distances descending with the row index (every accept list overflows its capacity: the redo
    kernel), sets smaller than k, ragged query counts, a radius bound; one and two queries per scanning wave."""
    from ucoslam_cv3_amd.knn import Index

    monkeypatch.setenv("UH_KNN_FORM", form)
    monkeypatch.setenv("UH_KNN_ACCEPT_QPW", accept_qpw)
    for case, (train, q) in _gpu_cases().items():
        index = Index(hip_ctx).build(train)
        for nn in range(1, 17):
            for s in (0, 1):
                idx, dist = index.search(q, nn, sorted=bool(s))
                ri, rd = oracle_lib.knn_search(oracle, train, q, nn, s)
                np.testing.assert_array_equal(idx, ri, err_msg=f"{case} nn={nn} sorted={s}")
                np.testing.assert_array_equal(dist, rd)
    train, q = synth.match_set(1003, 9999, seed=78)
    index = Index(hip_ctx).build(train)
    for nn, md in ((10, -1), (10, 70), (2, -1), (2, 55), (16, 90)):
        idx, dist = index.search(q, nn, sorted=True, max_dist=md)
        ri, rd = oracle_lib.knn_search(oracle, train, q, nn, 1, max_dist=md)
        np.testing.assert_array_equal(idx, ri, err_msg=f"nn={nn} max_dist={md}")
        np.testing.assert_array_equal(dist, rd)
    # a long descending run: row i is closer to every query than row i-1 -> every row is accepted, every list overflows
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    n = 700
    train = np.repeat(base[None, :], n, 0)
    bits = np.unpackbits(train, axis=1)
    for i in range(n):
        bits[i, : max(0, 250 - i // 3)] ^= 1
    train = np.packbits(bits, axis=1)
    q = np.repeat(base[None, :], 70, 0)
    index = Index(hip_ctx).build(train)
    for nn in (3, 10):
        idx, dist = index.search(q, nn, sorted=False)
        ri, rd = oracle_lib.knn_search(oracle, train, q, nn, 0)
        np.testing.assert_array_equal(idx, ri)
        np.testing.assert_array_equal(dist, rd)


@pytest.mark.gpu
def test_hip_knn_stream_form_many_workgroups_and_tag_wrap(hip_ctx, oracle, monkeypatch):
    """The one-launch form at the size it is the default for (8000+ queries: 32 replay workgroups beside 1000 scanning ones), repeated
    launches on one index with changing query counts (stale words of earlier launches carry other tags), and the launch tag running over
    2^32 (the list buffer is cleared and the tags start over)."""
    from ucoslam_cv3_amd.knn import Index

    monkeypatch.setenv("UH_KNN_TAG0", str(0xFFFFFFFC))
    train, q = synth.match_set(8003, 3001, seed=91)
    index = Index(hip_ctx).build(train)
    for rep, (nq, nn) in enumerate(((8003, 10), (7001, 10), (8003, 2), (6500, 16), (8003, 10), (8000, 5))):
        idx, dist = index.search(q[:nq], nn, sorted=bool(rep & 1))
        ri, rd = oracle_lib.knn_search(oracle, train, q[:nq], nn, rep & 1)
        np.testing.assert_array_equal(idx, ri, err_msg=f"launch {rep}: nq={nq} nn={nn}")
        np.testing.assert_array_equal(dist, rd)


@pytest.mark.gpu
def test_hip_knn_large_map_and_widest_rows(hip_ctx, oracle):
    """150 000 train rows (a large map), nn = 64 (the widest row the wave heap holds) and nn = 1, sorted and unsorted; a train
    count that is not a multiple of the 256-row scan group."""
    from ucoslam_cv3_amd.knn import Index

    train, q = synth.match_set(300, 150003, seed=41)
    index = Index(hip_ctx).build(train)
    for nn in (64, 1, 17):
        for s in (0, 1):
            idx, dist = index.search(q, nn, sorted=bool(s))
            ri, rd = oracle_lib.knn_search(oracle, train, q, nn, s)
            np.testing.assert_array_equal(idx, ri, err_msg=f"nn={nn} sorted={s}")
            np.testing.assert_array_equal(dist, rd)


@pytest.mark.gpu
def test_hip_knn_full_size_properties(hip_ctx, oracle):
    """BASELINE size (2000 x 10000): oracle on a query sample + size-independent properties on all rows."""
    import torch

    from ucoslam_cv3_amd.knn import Index

    train, q = synth.match_set(2000, 10000, seed=0)
    dt = torch.from_numpy(train).cuda()
    dq = torch.from_numpy(q).cuda()
    index = Index(hip_ctx).build(dt)
    for nn in (2, 10):
        idx, dist = index.search(dq, nn, sorted=True)
        torch.cuda.synchronize()
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        sample = np.arange(0, 2000, 16)
        ri, rd = oracle_lib.knn_search(oracle, train, q[sample], nn, 1)
        np.testing.assert_array_equal(idx[sample], ri)
        np.testing.assert_array_equal(dist[sample], rd)
        # sortedness + distances recomputed independently + each row's indices distinct
        assert (np.diff(dist, axis=1) >= 0).all()
        xor = np.bitwise_xor(q[:, None, :], train[idx])
        recomputed = np.unpackbits(xor, axis=2).sum(axis=2)
        np.testing.assert_array_equal(recomputed, dist)
        assert all(len(set(r)) == nn for r in idx)
        # the k-th distance bounds every other row's distance from below (checked on a row sample)
        for r in sample[:20]:
            alld = np.unpackbits(np.bitwise_xor(q[r][None, :], train), axis=1).sum(axis=1)
            assert np.sort(alld)[:nn].tolist() == dist[r].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("nshards,cap", [(2, 64), (8, 64), (3, 4)])
def test_hip_knn_sharded_replay_equals_single(hip_ctx, oracle, nshards, cap):
    """scan per shard -> concatenate candidate lists -> replay == unsharded reference (incl. cap overflow rescans)."""
    import torch

    from ucoslam_cv3_amd.knn import Index, shard_bounds

    for train, q in (synth.match_set(130, 1200, seed=41), synth.tie_stress_set(70, 900, seed=42), synth.descending_set(4, 500, seed=43)):
        dt, dq = torch.from_numpy(train).cuda(), torch.from_numpy(q).cuda()
        index = Index(hip_ctx).build(dt)
        b = shard_bounds(len(train), nshards)
        for nn, s in [(2, 1), (10, 0)]:
            cands, counts = [], []
            for sh in range(nshards):
                index.set_shard(b[sh], b[sh + 1])
                c, n = index.scan_shard(dq, nn, cap)
                cands.append(c)
                counts.append(n)
            index.set_shard(0, len(train))
            idx, dist = index.replay(dq, nn, torch.stack(cands), torch.stack(counts), sorted=bool(s))
            torch.cuda.synchronize()
            ri, rd = oracle_lib.knn_search(oracle, train, q, nn, s)
            np.testing.assert_array_equal(idx.cpu().numpy(), ri)
            np.testing.assert_array_equal(dist.cpu().numpy(), rd)


@pytest.mark.gpu
@pytest.mark.parametrize("nn", [10, 20])
def test_hip_knn_shard_scan_ignores_rows_behind_the_device_count(hip_ctx, oracle, nn):
    """A frame block of fixed capacity whose first `count` rows (a device-side number) are this frame's descriptors: the rows behind
    them emit EMPTY lists — stale descriptors cannot overflow a list (fstream.hip) — and the valid rows' results are those
    of the reference search."""
    import torch

    from ucoslam_cv3_amd.knn import Index

    train, q = synth.match_set(600, 1500, seed=77)
    valid = 250
    dt, dq = torch.from_numpy(train).cuda(), torch.from_numpy(q).cuda()
    index = Index(hip_ctx).build(dt)
    cnt = torch.tensor([valid], dtype=torch.int32, device="cuda")
    cap = 1024
    index.set_valid_rows(cnt)
    cand, counts = index.scan_shard(dq, nn, cap)
    torch.cuda.synchronize()
    c = counts.cpu().numpy()
    assert (c[valid:] == 0).all() and (c[:valid] > 0).all()
    idx, dist = index.replay(dq, nn, cand[None], counts[None])
    torch.cuda.synchronize()
    ri, rd = oracle_lib.knn_search(oracle, train, q[:valid], nn, 0)
    np.testing.assert_array_equal(idx.cpu().numpy()[:valid], ri)
    np.testing.assert_array_equal(dist.cpu().numpy()[:valid], rd)
    index.set_valid_rows(None)
    _, counts2 = index.scan_shard(dq, nn, cap)
    assert (counts2.cpu().numpy() > 0).all()


@pytest.mark.gpu
def test_hip_knn_error_behaviour(hip_ctx):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    index = Index(hip_ctx)
    q = np.zeros((3, 32), np.uint8)
    with pytest.raises(u.UcoslamHipError) as e:   # index.cpp:82-85: search on an unbuilt index fails
        index.search(q, 2)
    assert e.value.code == -3
    index.build(np.zeros((0, 32), np.uint8))      # index.cpp:49: empty features leave it unbuilt
    with pytest.raises(u.UcoslamHipError):
        index.search(q, 2)
    with pytest.raises(u.UcoslamHipError):        # only 32-byte descriptors
        index.build(np.zeros((4, 61), np.uint8))


@pytest.mark.gpu
def test_hip_knn_at_the_bench_launch_shape(hip_ctx, oracle):
    """Exactly the launch bench.py times: 4 x 2000 = 8000 query rows against a 10 000-row map, nn = 10, unsorted heap rows, device
    buffers in and out through uh_knn_search_dev (the one-launch streaming form at this size)."""
    import torch

    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd._lib import check, dev_ptr
    from ucoslam_cv3_amd.knn import Index

    train, q0 = synth.match_set(2000, 10000, seed=50)
    q = np.concatenate([q0, synth.match_set(2000, 10000, seed=51)[1], synth.match_set(2000, 10000, seed=52)[1], q0[::-1]])
    assert q.shape == (8000, 32)
    index = Index(hip_ctx).build(torch.from_numpy(train).cuda())
    dq = torch.from_numpy(q).cuda()
    idx = torch.empty((8000, 10), dtype=torch.int32, device="cuda")
    dist = torch.empty_like(idx)
    for _ in range(2):   # twice: the second launch runs on the first one's tagged lists
        check(u.lib().uh_knn_search_dev(index._h, dev_ptr(dq), 8000, 10, dev_ptr(idx), dev_ptr(dist), 0, -1))
        torch.cuda.synchronize()
        ri, rd = oracle_lib.knn_search(oracle, train, q, 10, 0)
        np.testing.assert_array_equal(idx.cpu().numpy(), ri)
        np.testing.assert_array_equal(dist.cpu().numpy(), rd)


@pytest.mark.gpu
def test_hip_knn_stream_form_overflow_then_regrown_lists(hip_ctx, oracle, monkeypatch):
    """the one-launch form picks its overflow counter by tag parity and clears only the NEXT launch's slot.  A launch with an
    odd tag whose lists all overflow (distances descending with the row index -> knn_redo_kernel), followed by a larger query set (the
    list buffer is reallocated, the tag starts over at 1 = odd again): the stale count must not be replayed."""
    from ucoslam_cv3_amd.knn import Index

    monkeypatch.setenv("UH_KNN_FORM", "stream")
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    n = 700
    bits = np.unpackbits(np.repeat(base[None, :], n, 0), axis=1)
    for i in range(n):
        bits[i, : max(0, 250 - i // 3)] ^= 1
    train = np.packbits(bits, axis=1)
    index = Index(hip_ctx).build(train)
    q_small = np.repeat(base[None, :], 90, 0)
    for rep in range(3):   # tags 1, 2, 3: the last overflowing launch has an odd tag
        idx, dist = index.search(q_small, 10, sorted=False)
        ri, rd = oracle_lib.knn_search(oracle, train, q_small, 10, 0)
        np.testing.assert_array_equal(idx, ri)
        np.testing.assert_array_equal(dist, rd)
    _, q_big = synth.match_set(4000, n, seed=7)   # 44x the queries: list_buf regrows, the tag is reset; random rows: nothing overflows
    idx, dist = index.search(q_big, 10, sorted=False)
    ri, rd = oracle_lib.knn_search(oracle, train, q_big, 10, 0)
    np.testing.assert_array_equal(idx, ri)
    np.testing.assert_array_equal(dist, rd)


@pytest.mark.gpu
def test_hip_knn_host_api_with_pinned_buffers(hip_ctx, oracle):
    """uh_knn_search with queries and result rows in pinned host memory (16-byte-wide copy launches + a polled completion word instead of
    copy-engine transfers and a stream synchronisation) returns the oracle's rows, also when only one side is pinned."""
    import ctypes as C

    import torch

    from ucoslam_cv3_amd._lib import check, lib, np_ptr
    from ucoslam_cv3_amd.knn import Index

    train, q = synth.match_set(2000, 10000, seed=21)
    index = Index(hip_ctx).build(torch.from_numpy(train).cuda())
    for nn, s in [(10, 0), (2, 1), (3, 0)]:     # (3: the row block is not a multiple of 16 bytes for odd query counts -> runtime copies)
        nq = 2000 if nn != 3 else 1999
        ref_i, ref_d = oracle_lib.knn_search(oracle, train, q[:nq], nn, s)
        pq = torch.from_numpy(q[:nq].copy()).pin_memory()
        pi = torch.zeros((nq, nn), dtype=torch.int32).pin_memory()
        pd = torch.zeros((nq, nn), dtype=torch.int32).pin_memory()
        for _ in range(2):
            check(lib().uh_knn_search(index._h, C.c_void_p(pq.data_ptr()), nq, 32, nn, C.c_void_p(pi.data_ptr()), C.c_void_p(pd.data_ptr()), s, -1))
        np.testing.assert_array_equal(pi.numpy(), ref_i)
        np.testing.assert_array_equal(pd.numpy(), ref_d)
        i2 = np.zeros((nq, nn), np.int32); d2 = np.zeros((nq, nn), np.int32)
        check(lib().uh_knn_search(index._h, C.c_void_p(pq.data_ptr()), nq, 32, nn, np_ptr(i2), np_ptr(d2), s, -1))   # pinned in, pageable out
        np.testing.assert_array_equal(i2, ref_i)
        np.testing.assert_array_equal(d2, ref_d)


@pytest.mark.gpu
def test_hip_knn_range_split_form_is_exact(hip_ctx, oracle, monkeypatch):
    """UH_KNN_FORM=split: a frame's queries against S slices of the train range (the sharded search's scan kernel, one launch) + the
    lane-per-query replay over the slices' exact accept lists — measured slower than the fused search and not the default, but it is the
    single-GPU rehearsal of the sharded path's kernels: rows must equal the oracle's, overflowed lists included (redo)."""
    import torch

    from ucoslam_cv3_amd.knn import Index

    monkeypatch.setenv("UH_KNN_FORM", "split")
    for (train, q) in (synth.match_set(700, 6000, seed=61), synth.tie_stress_set(300, 4000, seed=62), synth.descending_set(6, 3000, seed=63)):
        index = Index(hip_ctx).build(torch.from_numpy(train).cuda())
        dq = torch.from_numpy(q).cuda()
        for nn, s in [(10, 0), (2, 1), (16, 0)]:
            idx, dist = index.search(dq, nn, sorted=bool(s))
            torch.cuda.synchronize()
            ri, rd = oracle_lib.knn_search(oracle, train, q, nn, s)
            np.testing.assert_array_equal(idx.cpu().numpy(), ri)
            np.testing.assert_array_equal(dist.cpu().numpy(), rd)


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nn", [(90, 10), (2000, 10), (2000, 2), (5000, 10)])
def test_hip_knn_host_form_with_pinned_arrays_hands_the_rows_over_in_the_launch(hip_ctx, oracle, nq, nn):
    """uh_knn_search with PINNED result arrays (what bench.py's single_frame_latency and a tracker's host call use): the stream form's replay
    workgroups copy the rows to the host arrays and post the completion word themselves (round 5, knn_stream_kernel's KnnHostOut) — the
    one-query-per-wave form (<= ~2000 queries), the two-queries-per-wave form (5000) and a set whose every list OVERFLOWS (distances
    descending with the row index: the redone rows must be in the arrays before any workgroup copies).  Three calls each: the word sequence
    advances, the arrays are overwritten."""
    import ctypes as C
    import torch
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    L = u.lib()
    sets = [synth.match_set(nq, 3000, seed=11), synth.tie_stress_set(nq, 2000, seed=12)]
    if nq <= 128:
        sets.append(synth.descending_set(nq, 700, seed=13))
    for train, q in sets:
        index = Index(hip_ctx).build(train)
        hq = torch.from_numpy(np.ascontiguousarray(q)).pin_memory()
        hi = torch.full((nq, nn), -7, dtype=torch.int32).pin_memory()
        hd = torch.full((nq, nn), -7, dtype=torch.int32).pin_memory()
        ri, rd = oracle_lib.knn_search(oracle, train, q, nn, 0)
        for rep in range(3):
            hi.fill_(-7); hd.fill_(-7)
            from ucoslam_cv3_amd._lib import check as _check
            _check(L.uh_knn_search(index._h, C.c_void_p(hq.data_ptr()), nq, 32, nn, C.c_void_p(hi.data_ptr()), C.c_void_p(hd.data_ptr()), 0, -1))
            np.testing.assert_array_equal(hi.numpy(), ri)
            np.testing.assert_array_equal(hd.numpy(), rd)


@pytest.mark.gpu
def test_hip_knn_very_large_batches_leave_the_one_launch_form(hip_ctx, oracle):
    """The stream form's replay workgroups wait for each other (shared redo, host hand-over), so the form is used only while all of them fit
    the device at once (<= 2 per compute unit: 32 768 queries on MI355X); 40 000 queries take the two launches — same rows."""
    from ucoslam_cv3_amd.knn import Index

    train, _ = synth.match_set(16, 400, seed=21)
    q = np.concatenate([synth.match_set(2000, 400, seed=100 + s)[1] for s in range(20)])
    index = Index(hip_ctx).build(train)
    idx, dist = index.search(q, 10, sorted=False)
    ri, rd = oracle_lib.knn_search(oracle, train, q, 10, 0)
    np.testing.assert_array_equal(idx, ri)
    np.testing.assert_array_equal(dist, rd)
