"""Per-rank time of the pyramid-level shards (parallel.level_ranges) of ONE 1241x376 frame, next to the full extraction:
what each rank of a 2/4/8-GPU sharded extraction would spend before the all-gather."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd import parallel
from ucoslam_cv3_amd.orb import ORBextractor, FeatParams
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
ext = ORBextractor.create(ctx)
fp = FeatParams(2000, 8, 1.2)
frame = torch.from_numpy(synth.frame(1241, 376, seed=0)).cuda()[None]
def t(first, end):
    ext.setLevelRange(first, end)
    out = ext.extract_batch(frame, fp)
    for _ in range(5): ext.extract_batch(frame, fp, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ext.extract_batch(frame, fp, out)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 50, int(out[2][0])
full, n = t(0, -1)
print(f"full: {full:.1f} us, {n} keypoints")
for world in (2, 4, 8):
    r = parallel.level_ranges(1241, 376, 8, 1.2, world)
    ts = [t(a, b) for a, b in r]
    print(f"world {world}: ranges {r} -> us per rank {[round(x[0], 1) for x in ts]} (max {max(x[0] for x in ts):.1f}), rows {[x[1] for x in ts]}")
