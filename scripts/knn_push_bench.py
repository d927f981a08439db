import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, numpy as np
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index
torch.cuda.set_device(0)
ctx = u.Context(0, 0)
idx = Index(ctx)
L = u.lib()
L.uh_knn_debug_push_cycles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
for k in (2, 10, 32):
    out = np.zeros(4, np.int64)
    for _ in range(2):
        L.uh_knn_debug_push_cycles(idx._h, k, 2000, out.ctypes.data_as(C.c_void_p))
    print(f"k={k}: vector push {out[0]/2000:.0f} cycles, through feed_step {out[1]/2000:.0f}, scalar-path push {out[2]/2000:.0f} cycles")
