#!/usr/bin/env python
"""Sharded exact search on ONE GPU (every tile scanned here, the lists replayed as the all-gather would leave them): time of the tile scans
and of the replay by number of tiles, for the lane-per-query kernels (default, k <= 16) and the wave-per-query ones (UH_KNN_SHARD_FORM=old)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index, shard_bounds

ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
train, q = synth.match_set(2000, 10000, seed=5)
dq = torch.from_numpy(q).cuda()
NN = 10
for world in (1, 2, 4, 8):
    b = shard_bounds(len(train), world)
    rows = b[1] - b[0]
    exp = NN * (1 + np.log(rows / NN)); cap = int(-(-(exp + 4 * np.sqrt(exp)) // 16) * 16)
    tiles = [Index(ctx).build(torch.from_numpy(train[b[s]:b[s + 1]].copy()).cuda()).set_row_offset(b[s]) for s in range(world)]
    def scans():
        return [t.scan_shard(dq, NN, cap) for t in tiles]
    lists = scans()
    cand = torch.stack([c for c, _ in lists]); cnt = torch.stack([n for _, n in lists])
    def replay():
        return tiles[0].replay_tiles(dq, NN, cand, cnt)
    for fn in (scans, replay):
        for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): tiles[0].scan_shard(dq, NN, cap)
    torch.cuda.synchronize(); t_scan = (time.perf_counter() - t) / 20 * 1e6
    t = time.perf_counter()
    for _ in range(20): replay()
    torch.cuda.synchronize(); t_rep = (time.perf_counter() - t) / 20 * 1e6
    print(f"world {world}: cap {cap}, mean list {float(cnt.float().mean()):.1f}, one tile scan {t_scan:.1f} us, replay of {world} lists {t_rep:.1f} us, overflow {int(replay()[2])}")
