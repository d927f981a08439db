"""examples/track_stream.cpp: a C++ host over the C ABI running the tracker / mapper loop of one keyframe interval (host in, host
out, a fresh local BA per step through setParams / optimize / getResults) with no Python in the path."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "ucoslam-cv3_amd")


def _build(tmp_path):
    exe = str(tmp_path / "track_stream")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "examples", "track_stream.cpp"), "-L", LIBDIR, "-lucoslam_hip",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return exe


def test_cpp_host_program_compiles_and_has_no_cpu_path(tmp_path):
    import torch

    exe = _build(tmp_path)
    if not torch.cuda.is_available():
        out = subprocess.run([exe, "2", "1"], capture_output=True, text=True)
        assert out.returncode == 0 and "no device" in out.stdout and "no CPU path" in out.stdout


@pytest.mark.gpu
def test_cpp_host_program_runs_the_keyframe_loop(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "5", "3"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["keypoints"] == [2000, 2000, 2000, 2000]          # the synthetic scene fills the feature budget
    assert r["ba_iters"] == [5, 10] and r["ba_form"] == 1 and r["ba_lanes"] == 8
    assert r["first_row"][0] >= 0 and 0 <= r["first_row"][1] <= 256
    assert 0.2 < r["ms_per_step"] < 5.0 and r["ba_set_problem_ms"] < 0.2 and r["ba_get_results_ms"] < 0.1


def _build_tracker(tmp_path):
    exe = str(tmp_path / "tracker_frame")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "examples", "tracker_frame.cpp"), "-L", LIBDIR, "-lucoslam_hip",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return exe


def test_tracker_chain_host_compiles_and_has_no_cpu_path(tmp_path):
    import torch

    exe = _build_tracker(tmp_path)
    if not torch.cuda.is_available():
        out = subprocess.run([exe, "2", "1"], capture_output=True, text=True)
        assert out.returncode == 0 and "no device" in out.stdout and "no CPU path" in out.stdout


@pytest.mark.gpu
def test_tracker_chain_host_runs_the_per_frame_sequence(tmp_path):
    """examples/tracker_frame.cpp: ORB -> kd-tree -> previous-frame projection search -> PnP -> map projection search -> PnP, one frame at
    a time, host buffers in and out of every call.  The scene is made from the frame's own features with a known pose, so the chain is
    checked end to end: the searches find their points, the solver keeps them and lands on the ground-truth pose."""
    exe = _build_tracker(tmp_path)
    out = subprocess.run([exe, "40", "5"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["keypoints"] == 2000.0 and r["prev_items"] == 800 and r["map_points"] == 3000
    assert r["matches_prev"] > 700 and r["matches_map"] > 500               # the projected points find their keypoints
    assert r["inliers1"] > 0.9 * r["matches_prev"] and r["inliers2"] > 1000  # ... and the pose-only solves keep them
    assert r["max_pose_err_vs_truth"] < 0.01                                 # float 3x4 entries against the ground-truth pose
    for k in ("orb_extract_ms", "set_frame_ms", "match_prev_ms", "pnp1_ms", "match_map_ms", "pnp2_ms"):
        assert 0.005 < r[k] < 2.0, (k, r[k])
    assert r["tracker_frame_ms"] < 3.0 and "host" in r["frame_route"]
    # the device route (the frame stays in HBM, the kd-tree is built there by kdbuild.hpp) finds exactly the same matches and pose
    out = subprocess.run([exe, "40", "5", "dev"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    h = json.loads(out.stdout.strip().splitlines()[-1])
    assert "device" in h["frame_route"]
    for k in ("keypoints", "matches_prev", "matches_map", "inliers1", "inliers2", "max_pose_err_vs_truth"):
        assert h[k] == r[k], (k, h[k], r[k])
    # ... and so does the resident frame whose tree the host core builds (uh_dev_frame_set_tree_builder(frame, 1)), with the four tracker
    # calls one after the other and as ONE call (uh_track_pose: list handling and look-ups on the device)
    for route, tag in (("devhost", "host core"), ("fused", "uh_track_pose")):
        out = subprocess.run([exe, "40", "5", route], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        h = json.loads(out.stdout.strip().splitlines()[-1])
        assert tag in h["frame_route"]
        for k in ("keypoints", "matches_prev", "matches_map", "inliers1", "inliers2", "max_pose_err_vs_truth"):
            assert h[k] == r[k], (route, k, h[k], r[k])
