"""One small pass of the hot path on cuda:0, checked against the CPU oracle (called by __graft_entry__.smoke())."""
import numpy as np
import torch

import oracle_lib
import synth


def run():
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    torch.cuda.set_device(0)
    ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
    L = oracle_lib.load_oracle()
    # --- Hamming kNN
    train, q = synth.match_set(128, 1000, seed=7)
    index = Index(ctx).build(torch.from_numpy(train).cuda())
    idx, dist = index.search(torch.from_numpy(q).cuda(), 2, sorted=True)
    torch.cuda.synchronize()
    ri, rd = oracle_lib.knn_search(L, train, q, 2, 1)
    assert (idx.cpu().numpy() == ri).all() and (dist.cpu().numpy() == rd).all(), "kNN mismatch vs oracle"
    for hook in _EXTRA:
        hook(ctx, L)


_EXTRA = []
