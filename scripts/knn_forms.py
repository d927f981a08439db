"""HIP-event timing of the exact kNN forms (UH_KNN_FORM=fused|twophase|stream, or the default policy) over k and the query count,
10 000 train rows; the table behind the form policy in csrc/knn.hip (uh_knn::two_phase_min_nq / stream_min_nn)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
train, q = synth.match_set(2000, 10000, seed=0)
index = Index(ctx).build(torch.from_numpy(train).cuda())
qq = np.concatenate([synth.match_set(2000, 10000, seed=s)[1] for s in range(6)])
def t(nq, nn):
    dq = torch.from_numpy(qq[:nq]).cuda()
    for _ in range(5): index.search(dq, nn, sorted=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): index.search(dq, nn, sorted=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1000
print(os.environ.get("UH_KNN_FORM"), "nq=8000:", {nn: round(t(8000, nn), 1) for nn in (1, 2, 3, 4, 5, 6, 8, 10, 16)})
print(os.environ.get("UH_KNN_FORM"), "nn=10:", {nq: round(t(nq, 10), 1) for nq in (2000, 3000, 4000, 5000, 6000, 12000)})
print(os.environ.get("UH_KNN_FORM"), "nn=2:", {nq: round(t(nq, 2), 1) for nq in (2000, 4000, 6000, 12000)})
