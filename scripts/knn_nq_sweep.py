"""Exact search, 10 000 train rows: microseconds per search (HIP events) against the number of queries, for the form in UH_KNN_FORM.
usage: python scripts/knn_nq_sweep.py [nn = 10]"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
train, q = synth.match_set(2000, 10000, seed=0)
qq = np.concatenate([synth.match_set(2000, 10000, seed=s)[1] for s in range(4)])
index = Index(ctx).build(torch.from_numpy(train).cuda())
NN = int(sys.argv[1]) if len(sys.argv) > 1 else 10
out = {}
for nq in (64, 256, 512, 1000, 1500, 2000, 2500, 3000, 4000, 8000):
    dq = torch.from_numpy(qq[:nq]).cuda()
    for _ in range(5): index.search(dq, NN, sorted=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): index.search(dq, NN, sorted=False)
    e1.record(); torch.cuda.synchronize()
    out[nq] = round(e0.elapsed_time(e1) / 40 * 1000, 1)
print(os.environ.get("UH_KNN_FORM", "default"), f"nn={NN}", out)
