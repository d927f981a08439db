"""One frame's exact search (2000 x 10 000, nn 10) in each form, for rocprofv3 --kernel-trace --stats: UH_KNN_FORM=fused|twophase|stream."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
train, q = synth.match_set(2000, 10000, seed=0)
qq = np.concatenate([synth.match_set(2000, 10000, seed=s)[1] for s in range((nq + 1999) // 2000)])[:nq]
index = Index(ctx).build(torch.from_numpy(train).cuda())
dq = torch.from_numpy(qq).cuda()
for _ in range(5): index.search(dq, 10, sorted=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): index.search(dq, 10, sorted=False)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("UH_KNN_FORM"), nq, "us per search:", round(e0.elapsed_time(e1) / 50 * 1000, 1))
