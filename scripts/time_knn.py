"""HIP-event timing of the exact kNN kernel at the BASELINE size (2000 and 8000 queries x 10000 x 32 B) with a result checksum."""
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
dt = torch.from_numpy(train).cuda()
index = Index(ctx).build(dt)
for nq in (2000, 8000):
    qq = np.concatenate([synth.match_set(2000, 10000, seed=s)[1] for s in range(nq // 2000)])
    dq = torch.from_numpy(qq).cuda()
    for nn, srt in ((2, True), (10, False)):
        for _ in range(5): r = index.search(dq, nn, sorted=srt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): index.search(dq, nn, sorted=srt)
        e1.record(); torch.cuda.synchronize()
        chk = int(r[0].sum().item()) ^ int(r[1].sum().item())
        print(f"nq={nq} nn={nn}: {e0.elapsed_time(e1)/30*1000:.1f} us per search  checksum {chk}")

# per-kernel split of the two-phase search (HIP events on the context stream) and the queries-per-wave forms
for qpw in (1, 2, 4):
    index.set_queries_per_wave(qpw)
    for _ in range(3): index.search(dq, 10, sorted=False)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(20): index.search(dq, 10, sorted=False)
    torch.cuda.synchronize()
    rep = ctx.prof_report(); ctx.prof_enable(False)
    print(f"qpw={qpw} nq=8000 nn=10:", {k.split('::')[-1]: round(1e3 * v[1] / v[0], 1) for k, v in rep.items() if v[0]}, "us per launch")
