"""The drop-in boundary: the C-ABI library loads and exports every symbol include/ucoslam_hip.h declares; the C++ adaptors
compile against it; without a GPU every entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "ucoslam_hip.h")
LIB = os.path.join(ROOT, "ucoslam-cv3_amd", "libucoslam_hip.so")


def _declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "HIP library not built: run __graft_entry__.build()"
    names = _declared_functions()
    assert len(names) >= 40
    L = C.CDLL(LIB)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in ucoslam_hip.h but not exported: {missing}"


def test_struct_layouts_match_reference_types():
    import numpy as np

    from ucoslam_cv3_amd.matcher import DMATCH_DTYPE
    from ucoslam_cv3_amd.orb import KEYPOINT_DTYPE, FeatParams

    assert KEYPOINT_DTYPE.itemsize == 28          # cv::KeyPoint: pt(2f) size angle response octave class_id
    assert DMATCH_DTYPE.itemsize == 16            # cv::DMatch: queryIdx trainIdx imgIdx distance
    assert C.sizeof(FeatParams) == 20             # Feature2DSerializable::FeatParams (dumped raw by ORBextractor::toStream_impl)
    fp = FeatParams()
    assert (fp.maxFeatures, fp.nOctaveLevels, fp.nthreads) == (4000, 8, -1) and abs(fp.scaleFactor - 1.2) < 1e-7


def _build_adaptor_probe(tmp_path):
    exe = str(tmp_path / "adaptors_probe")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host_helpers", "adaptors_compile.cpp"),
                           "-L", os.path.dirname(LIB), "-lucoslam_hip", f"-Wl,-rpath,{os.path.dirname(LIB)}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_adaptors_compile_and_fail_loudly_without_gpu(tmp_path):
    import torch

    exe = _build_adaptor_probe(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    if not torch.cuda.is_available():
        assert "no device" in out.stdout and "no CPU path" in out.stdout


@pytest.mark.gpu
def test_cpp_adaptors_run_on_gpu(tmp_path):
    exe = _build_adaptor_probe(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "gpu path ok: first neighbour 0 dist 0" in out.stdout, out.stdout + out.stderr
