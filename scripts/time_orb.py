"""HIP-event timing of the ORB extractor on resident frames (1241x376, 2000 features), batch 1 and 8."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.orb import ORBextractor, FeatParams
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
ext = ORBextractor.create(ctx)
fp = FeatParams(2000, 8, 1.2)
for B in (1, 2, 4, 8):
    frames = torch.from_numpy(np.stack([synth.frame(1241, 376, seed=s % 8, shift=(2*s, s)) for s in range(B)])).cuda()
    out = ext.extract_batch(frames, fp)
    for _ in range(3): ext.extract_batch(frames, fp, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 20
    e0.record()
    for _ in range(N): ext.extract_batch(frames, fp, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / N
    print(f"orb batch={B}: {ms*1000:.1f} us per launch-set, {ms*1000/B:.1f} us/frame, counts={out[2].cpu().numpy()[:4]}")
for B in (1, 2, 4):
    frames = torch.from_numpy(np.stack([synth.frame(1241, 376, seed=s % 8, shift=(2*s, s)) for s in range(B)])).cuda()
    out = ext.extract_batch(frames, fp)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(10): ext.extract_batch(frames, fp, out)
    rep = ctx.prof_report(); ctx.prof_enable(False)
    print(f"B={B}", {k.split('::')[-1]: round(1e3*v[1]/v[0], 1) for k, v in rep.items() if v[0]})
