"""flatten_for_ba (include/ucoslam_hip/flatten_ba.hpp): GlobalOptimizerG2O::setParams' graph-selection rules
(globaloptimizer_g2o.cpp:99-172, 191-249) as code, checked on a toy map against hand-derived vertex / edge sets — a point with one
observer dropped, extra observers joining as FIXED_WITHOUTPOINTS, fixFirstFrame / fixed_frames, the (double)(float)(1./scaleFactor)
information scalar, getResults' write-back.  Pure host C++, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flatten_for_ba_rules_on_a_toy_map(tmp_path):
    exe = str(tmp_path / "flatten_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "host_helpers", "flatten_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "flatten ok" in out.stdout, out.stdout + out.stderr
