"""MI355X-native tracking hot path of UcoSLAM (ORB extract -> Hamming kNN / BoW -> LM/Schur BA).

The product is the HIP library `libucoslam_hip.so` behind the C ABI of include/ucoslam_hip.h;
this package is the thin Python host side used by tests/, bench.py and __graft_entry__.py.
"""
from ._lib import Context, UcoslamHipError, lib  # noqa: F401
from . import ba, bow, knn, matcher, orb, parallel, pnp, projmatch, slm  # noqa: F401,E402  (registers the ctypes prototypes)
