import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
from ucoslam_cv3_amd.orb import ORBextractor, FeatParams
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
train, q_matched = synth.match_set(2000, 10000, seed=0)
rng = np.random.default_rng(1)
q_rand = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
ext = ORBextractor.create(ctx)
fr = torch.from_numpy(synth.frame(1241, 376, seed=0)[None]).cuda()
kps, desc, counts = ext.extract_batch(fr, FeatParams(2000, 8, 1.2)); torch.cuda.synchronize()
q_orb = desc[0].cpu().numpy()
print("orb desc: unique rows", len(np.unique(q_orb, axis=0)), "mean popcount", np.unpackbits(q_orb, axis=1).sum(1).mean())
dt = torch.from_numpy(train).cuda()
index = Index(ctx).build(dt)
def t(q, nn, s):
    dq = torch.from_numpy(q).cuda()
    for _ in range(3): index.search(dq, nn, sorted=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): index.search(dq, nn, sorted=s)
    e1.record(); torch.cuda.synchronize()
    c, n = index.scan_shard(dq, nn, 4096); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1000, float(n.float().mean()), int(n.max())
for name, q in (("matched", q_matched), ("random", q_rand), ("orb", q_orb)):
    for nn in (2, 10):
        us, mean_acc, max_acc = t(q, nn, False)
        print(f"{name:8s} nn={nn:2d}: {us:7.1f} us   accepted/query mean {mean_acc:.1f} max {max_acc}")
