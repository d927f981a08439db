"""Scan time of the accept kernel against the number of train rows (2000 queries, nn 10): the slope is the steady-state cost of a 256-row group,
the intercept the early steps' accept handling.  Run under rocprofv3 --kernel-trace --stats with UH_KNN_FORM=twophase."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
nt = int(sys.argv[1])
train, q = synth.match_set(2000, nt, seed=0)
index = Index(ctx).build(torch.from_numpy(train).cuda())
dq = torch.from_numpy(q).cuda()
for _ in range(30): index.search(dq, 10, sorted=False)
torch.cuda.synchronize()
