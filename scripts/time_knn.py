"""Quick HIP-event timing of the kNN kernel at the BASELINE size (2000 x 10000 x 32 B)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
train, q = synth.match_set(2000, 10000, seed=0)
dt, dq = torch.from_numpy(train).cuda(), torch.from_numpy(q).cuda()
index = Index(ctx).build(dt)
for nn in (2, 10):
    for _ in range(5): index.search(dq, nn, sorted=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): index.search(dq, nn, sorted=True)
    e1.record(); torch.cuda.synchronize()
    print(f"knn nn={nn}: {e0.elapsed_time(e1)/50*1000:.1f} us per search")
