"""What slows the local BA down when another stream is busy?  One local BA (own stream, worker thread) next to ONE background launch
of a chosen character on a second stream, sized to last about as long as the BA: sleeping waves (occupancy only), integer VALU,
streaming loads, LDS traffic — at 2 and 8 resident waves per SIMD."""
import sys, os, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd._lib import check, dev_ptr
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
L = u.lib()
L.uh_debug_background.restype = C.c_int
L.uh_debug_background.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
ct, cb = u.Context(0, private=True), u.Context(0, private=True)
ba = GlobalOptimizer.create(cb)
ba.setParams(synth.ba_problem(10, 3000, seed=0), ParamSet(nIters=5))
buf = torch.randint(0, 2**31 - 1, (64 * 1024 * 1024 // 4,), dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")

def bg(mode, blocks, iters):
    check(L.uh_debug_background(ct.handle, mode, blocks, iters, dev_ptr(buf), buf.numel() * 4, dev_ptr(sink)))

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n

def ba_only():
    ba.optimize_async(); ba.wait()
t_ba = timed(ba_only)
print(f"BA alone: {t_ba:.3f} ms")
names = {0: "sleeping waves", 1: "integer VALU", 2: "streaming loads", 3: "LDS traffic"}
for mode in (0, 1, 2, 3):
    for wps in (2, 8):
        blocks = 256 * wps          # 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
        # calibrate iters so that the background launch alone lasts ~0.8 ms
        iters = 200
        for _ in range(6):
            t = timed(lambda: bg(mode, blocks, iters), 5)
            iters = max(1, int(iters * 0.8 / max(t, 1e-3)))
        t_bg = timed(lambda: bg(mode, blocks, iters), 10)
        def both():
            ba.optimize_async(); bg(mode, blocks, iters); ba.wait()
        t_both = timed(both)
        print(f"{names[mode]:16s} {wps} waves/SIMD: background alone {t_bg:.3f} ms, BA next to it {t_both:.3f} ms (BA alone {t_ba:.3f})", flush=True)
