"""Soak of the tracking stream beside the local BA: ORB (4 frames) + exact kNN (8000 x 10000, nn 10 -> the one-launch search, and nn 2 -> the
two-launch form; round 5: also ONE frame's 2000 queries at nn 10 and nn 2 -> the one-query-per-wave form of the one-launch search, whose replay
workgroups wait for each other at the end) repeated while `optimize_async` runs on its own stream; every result is compared with the unloaded one.
usage: python scripts/soak_tracking.py [seconds]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.orb import ORBextractor, FeatParams
from ucoslam_cv3_amd.knn import Index
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
ctx_ba = u.Context(0, private=True)
frames = torch.from_numpy(np.stack([synth.frame(seed=i, shift=(3 * i, i)) for i in range(4)])).cuda()
ext = ORBextractor.create(ctx); fp = FeatParams(2000, 8, 1.2)
train, _ = synth.match_set(2000, 10000, seed=0)
index = Index(ctx).build(torch.from_numpy(train).cuda())
ba = GlobalOptimizer.create(ctx_ba); ba.setParams(synth.ba_problem(10, 3000, 0, nfixed=2), ParamSet(nIters=5))

def once():
    kps, desc, counts = ext.extract_batch(frames, fp)
    q = desc.reshape(-1, 32)
    i10, d10 = index.search(q, 10, sorted=False)
    i2, d2 = index.search(q, 2, sorted=True)
    j10, e10 = index.search(q[:2000], 10, sorted=False)
    j2, e2 = index.search(q[2000:4000], 2, sorted=False)
    torch.cuda.synchronize()
    return [t.clone() for t in (kps.view(torch.uint8), desc, counts, i10, d10, i2, d2, j10, e10, j2, e2)]

ref = once()
t0 = time.time(); n = bad = 0
while time.time() - t0 < budget:
    ba.optimize_async()
    got = once()
    ba.wait()
    n += 1
    if not all(torch.equal(a, b) for a, b in zip(ref, got)): bad += 1
print(f"{n} tracking steps beside the local BA, {bad} differing from the unloaded result")
