#!/usr/bin/env python
"""bench.py — tracking frames/sec of the MI355X hot path (ORB extract + Hamming match + local BA), 1241x376 mono.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run, one
rank per GPU.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1..3] combined = the metric's "ORB extract + match + local BA"):
  one step = `frames_per_step` (4) synthetic 1241x376 frames = one keyframe interval, THROUGH THE PLUGIN BOUNDARY (host in, host out):
     the 4 frames enter from pinned host memory (one async H2D on the tracking stream),
     ORB extraction of the 4 frames in one batched launch set (8 levels, scale 1.2, 2000 features)
     4 x brute-force Hamming kNN, 2000 query descriptors (the frame's own ORB output) vs a 10 000-descriptor map,
         nn=10 unsorted — the FrameMatcher_Flann call shape (framematcher.cpp:213,239)
     keypoints, descriptors, counts and the 4 x 2000 x 10 match rows return to pinned host memory (one async D2H),
     1 x local BA, 10 keyframes x 3000 landmarks (~26k observations), nIters=5 (+10), fp64 — a FRESH problem every step, the three
         phases of the plugin: setParams + optimize on the mapper thread (mapmanager.cpp:11388-11405), getResults on the tracker
         thread (:1267-1305); the flattened arrays live in pageable host memory, the results return to host arrays
  value = frames / second over all ranks (median of R repetitions of the K-step loop; min / max beside it).  The resident-loop figure
  of rounds 1-2 (frames already in HBM, one problem re-optimised, nothing copied back) is stages.kernel_only_frames_per_s.  Multi-GPU: frame streams are independent, so every rank runs the same per-GPU
  workload on its own frames ("weak" scaling, no data-path collective); only the timing reduction crosses ranks.  With N > 1 the
  line also carries stages.sharded_*: ONE frame stream over all N GPUs (BASELINE config 5) — pyramid levels and train tiles
  sharded, every rank holding only its tile, one fused RCCL all-gather per frame (ucoslam_cv3_amd.parallel.ShardedFrameStream).
Extra objects: "roofline" (dominant kernel, HIP events on its launch stream; SURVEY §8(d) bytes per unit x units per launch) and
"cpu_baseline" (rank 0, N=1: oracle ORB at 1 / 2 / all threads, real xflann at 1 / all threads, real g2o).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

W, H = 1241, 376
MAX_FEATURES, NLEVELS, SCALE = 2000, 8, 1.2
NQ, NT, NN = 2000, 10000, 10
BA_K, BA_P = 10, 3000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def level_sizes(w=None, h=None):
    w, h = w or W, h or H
    sc = np.float32(1.0)
    out = []
    for l in range(NLEVELS):
        out.append((int(np.rint(np.float32(w) * (np.float32(1.0) / sc))), int(np.rint(np.float32(h) * (np.float32(1.0) / sc)))))
        sc = np.float32(sc * np.float32(SCALE))
    return out


def survey_8d_bytes(n_kpts, ba_E, ba_P=BA_P, ba_K=BA_K, w=None, h=None):
    """SURVEY.md §8(d), verbatim: algorithmic bytes per unit of each stage of the path.
       B_orb   = W*H + 2*sum_l (w_l+38)(h_l+38) + N*(28+32)                    per frame
       B_match = (NQ+NT)*32 + NQ*k*8                                            per frame (k = 10: the FrameMatcher_Flann call shape)
       B_ba    = E*32 + 2*P*24 + 2*K*56 + K^2*288                               per LM iteration"""
    w, h = w or W, h or H
    lv = level_sizes(w, h)
    b_orb = w * h + 2 * sum((wl + 38) * (hl + 38) for wl, hl in lv) + n_kpts * 60
    b_match = (NQ + NT) * 32 + NQ * NN * 8
    b_ba = ba_E * 32 + 2 * ba_P * 24 + 2 * ba_K * 56 + ba_K * ba_K * 288
    return {"orb_per_frame": b_orb, "match_per_frame": b_match, "ba_per_lm_iteration": b_ba}


ORB_KERNELS = ("blur7_kernel", "copy_kernel", "resize_cubic_kernel", "fast_score_kernel", "cell_nms_kernel", "select_kernel", "describe_kernel",
               "nonmax_kernel")
MATCH_KERNELS = ("knn_search_kernel", "knn_search_mq_kernel", "knn_accept_kernel", "knn_replay_lane_kernel", "knn_stream_kernel", "knn_redo_kernel")


def persistent_ba_exchange_bytes(ba_P, trials):
    """What the persistent BA kernel moves BY DESIGN per launch (csrc/ba_persist.hpp): the observations once, then per LM trial the
    reduction partials between its workgroups (G partials of 1516 doubles written, read slice-wise, the reduced vector read by every
    workgroup, chi2 partials) — all of it write-through stores / L1-bypassing loads, served by L2 / MALL.  Every exchanged double
    travels as two tagged 64-bit words (16 bytes)."""
    lw = max(8, min(32, -(-ba_P // 64)))
    g = -(-ba_P // lw)
    nelem = 78 * 16 + 8 * 27 + 48 + 4
    per_trial = g * nelem * 16 * 2 + g * nelem * 16 + nelem * 16 + g * 64 * (1 + g)
    return int(ba_P * 8 * 28 + trials * per_trial), g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--knn-qpw", type=int, default=1, help="queries per wave of the exact matcher (1, 2, 4)")
    ap.add_argument("--quick", action="store_true", help="headline step only: skip the per-stage and side measurements")
    ap.add_argument("--reps", type=int, default=15, help="repetitions of the timed K-step loop (median / min / max are reported)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if "RANK" in os.environ:   # launched by torch.distributed.run (also with one rank, so that the path is exercised)
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import synth
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
    from ucoslam_cv3_amd.knn import Index
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    F = args.frames_per_step
    dev = torch.device("cuda", local_rank)
    # UH_BENCH_CU_SPLIT=<n>: the mapper's stream on mask bits [0, n) and the tracker's on [n, all) of the compute units (n / 8 CUs of every
    # XCD for the local BA, the rest for the tracker: uh_ctx_create_private_cus) — the two never share a CU
    cu_split = int(os.environ.get("UH_BENCH_CU_SPLIT", "0"))
    n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    if cu_split:
        _trk_ctx = u.Context(local_rank, cus=(cu_split, n_cus - cu_split))
        torch.cuda.set_stream(torch.cuda.ExternalStream(u.lib().uh_ctx_stream(_trk_ctx._h), device=dev))
    ctx = u.Context(local_rank, torch.cuda.current_stream().cuda_stream)

    # ---- synthetic inputs: frames in pinned host memory (the camera's buffers), the map's descriptors resident in HBM (the map lives on
    # the device between frames), N_PROB different local-BA problems as flattened arrays in pageable host memory
    frames_np = np.stack([synth.frame(W, H, seed=1000 * rank + f, shift=(2 * f, f)) for f in range(F)])
    frames_host = torch.from_numpy(frames_np).pin_memory()
    frames = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    frames.copy_(frames_host)
    map_desc_np, _ = synth.match_set(1, NT, seed=50 + rank)
    map_desc = torch.from_numpy(map_desc_np).to(dev)
    N_PROB = 4
    ba_problems = [synth.ba_problem(BA_K, BA_P, seed=rank * N_PROB + i) for i in range(N_PROB)]
    ba_pr = ba_problems[0]

    ext = ORBextractor.create(ctx)
    fp = FeatParams(MAX_FEATURES, NLEVELS, SCALE)
    # ONE device block for everything a step hands back to the host: keypoints | descriptors | match rows | counts -> one D2H copy
    o_kps, o_desc = 0, F * MAX_FEATURES * 28
    o_idx = o_desc + F * MAX_FEATURES * 32
    o_dist = o_idx + F * NQ * NN * 4
    o_cnt = o_dist + F * NQ * NN * 4
    out_bytes = o_cnt + 64
    out_dev = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
    out_host = torch.zeros(out_bytes, dtype=torch.uint8).pin_memory()
    kps_v = out_dev[o_kps:o_desc].view(torch.float32).view(F, MAX_FEATURES, 7)
    desc_v = out_dev[o_desc:o_idx].view(F, MAX_FEATURES, 32)
    knn_idx = out_dev[o_idx:o_dist].view(torch.int32).view(F, NQ, NN)
    knn_dist = out_dev[o_dist:o_cnt].view(torch.int32).view(F, NQ, NN)
    cnt_v = out_dev[o_cnt:o_cnt + 4 * F].view(torch.int32)
    orb_out = (kps_v, desc_v, cnt_v)
    ext.extract_batch(frames, fp, orb_out)
    index = Index(ctx).build(map_desc)
    # queries per wave of the exact matcher: round 1 used 4 (a quarter of the L1/L2 streaming beside the latency-bound BA launch chain);
    # with the local BA as one persistent launch the step is bound by that launch alone and the plain one-query form is the fastest
    # (0.740 / 0.750 / 0.751 ms per step at 1 / 2 / 4)
    index.set_queries_per_wave(args.knn_qpw)
    # the reference runs local BA on its mapper thread, concurrently with tracking (mapmanager.cpp:1550);
    # here BA gets its own HIP stream so that its kernel overlaps the tracking stream's
    ctx_ba = u.Context(local_rank, cus=(0, cu_split)) if cu_split else u.Context(local_rank, private=True)
    ba = GlobalOptimizer.create(ctx_ba).wantChi2(False)   # (GlobalOptimizer::getResults returns poses, points and bad associations: no chi2)
    ba_ps = ParamSet(nIters=5)
    L = u.lib()
    from ucoslam_cv3_amd._lib import check, dev_ptr, np_ptr
    ba_out = dict(poses=np.zeros((BA_K, 16), np.float32), points=np.zeros((BA_P, 3), np.float32), bad=np.zeros(max(p_["E"] for p_ in ba_problems), np.uint8),
                  iters=np.zeros(2, np.int32))
    step_no = [0]
    ba_prepared = [ba.prepareProblem(p_) for p_ in ba_problems]   # (ctypes views of the flattened arrays, built once)

    # One keyframe interval through the plugin boundary.  Mapper thread (the BA object's worker, uh_ba_solve_async): setParams on a
    # FRESH problem + optimize.  Tracker thread (this one): frames in, ORB, match, results out, then getResults of the mapper's BA.
    # (the C entry points with their argument tuples built once: the harness is Python, the path it times is not)
    import ctypes as C
    solve_args = [(ba._h, C.byref(pr_[0]), 0, 0, 0, C.byref(ba_ps), None) for pr_ in ba_prepared]
    get_args = (ba._h, np_ptr(ba_out["poses"]), np_ptr(ba_out["points"]), None, np_ptr(ba_out["bad"]), np_ptr(ba_out["iters"]))
    knn_args = (index._h, dev_ptr(desc_v), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1)
    cap_kp = max(L.uh_orb_max_keypoints(ext._h), 1)
    orb_args = (ext._h, dev_ptr(frames), W, H, W, W * H, F, dev_ptr(kps_v), dev_ptr(desc_v), cap_kp, dev_ptr(cnt_v))
    trk_stream = torch.cuda.current_stream()

    def step():
        i = step_no[0] % N_PROB
        step_no[0] += 1
        check(L.uh_ba_solve_async(*solve_args[i]))                  # mapper thread: setParams (fresh problem) + optimize
        frames.copy_(frames_host, non_blocking=True)                # tracker: 4 frames in
        check(L.uh_orb_extract_dev(*orb_args))
        # the F frames' descriptor blocks are contiguous [F, 2000, 32]: one launch matches all F x 2000 queries against the map
        check(L.uh_knn_search_dev(*knn_args))
        out_host.copy_(out_dev, non_blocking=True)                  # keypoints, descriptors, counts, match rows out
        trk_stream.synchronize()                                    # the tracker owns its host buffers again (its work ends long before the mapper's: waited for first,
                                                                    #   the stream synchronisation — ~8 us of runtime call even on an idle stream — is off the step's critical path)
        check(L.uh_ba_wait(ba._h))
        check(L.uh_ba_get_results(*get_args))                       # tracker thread: getResults of the mapper's BA

    # rounds 1-2's step, kept as stages.kernel_only_*: frames resident in HBM, ONE problem re-optimised, nothing returns to the host
    ba_res = GlobalOptimizer.create(ctx_ba).wantChi2(False)
    ba_res.setParams(ba_pr, ba_ps)

    def step_resident():
        ba_res.optimize_async()
        ext.extract_batch(frames, fp, orb_out)
        check(L.uh_knn_search_dev(index._h, dev_ptr(desc_v), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1))
        ba_res.wait()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    reps_s = []
    for _ in range(max(args.reps, 1)):   # R repetitions of EXACTLY K steps, each bracketed by barrier + synchronize
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        reps_s.append(time.perf_counter() - t0)
    if dist is not None:
        dist.barrier()
        tt = torch.tensor(reps_s, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)   # per repetition: the slowest rank
        reps_s = [float(v) for v in tt.tolist()]
    t_max = float(np.median(reps_s))
    counts = out_host[o_cnt:o_cnt + 4 * F].view(torch.int32).numpy().copy()   # (as the host received them)
    full_frames = bool((counts == MAX_FEATURES).all())   # the synthetic scene yields the full 2000-keypoint budget
    total_frames = world * F * args.steps
    value = total_frames / t_max
    ms_per_step = 1e3 * t_max / args.steps
    ms_min, ms_max = 1e3 * min(reps_s) / args.steps, 1e3 * max(reps_s) / args.steps

    # ---- per-stage split and roofline (rank 0; separate passes so event overhead never enters the headline number)
    roofline = None
    stage_ms = {}
    if rank == 0 and args.quick:
        print(json.dumps({"value": round(value, 1), "ms_per_step": round(ms_per_step, 4), "ms_per_step_min": round(ms_min, 4), "ms_per_step_max": round(ms_max, 4), "reps": len(reps_s), "knn_qpw": args.knn_qpw, "knn_form": os.environ.get("UH_KNN_FORM", "auto (nn >= 6: scan + replay in one launch)")}), flush=True)
        return
    if rank == 0:
        def timed(fn, reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / reps

        stage_ms["orb_ms_per_frame"] = timed(lambda: ext.extract_batch(frames, fp, orb_out), 20) / F
        stage_ms["match_ms_per_frame"] = timed(
            lambda: check(L.uh_knn_search_dev(index._h, dev_ptr(orb_out[1]), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1)), 50) / F
        stage_ms["match_ms_single_frame_launch"] = timed(
            lambda: check(L.uh_knn_search_dev(index._h, dev_ptr(orb_out[1][0]), NQ, NN, dev_ptr(knn_idx[0]), dev_ptr(knn_dist[0]), 0, -1)), 50)
        # the local BA by phase of the plugin protocol (this thread, nothing else on the GPU): a FRESH problem per call as in the step
        def ba_phases(reps):
            ts, to, tg = [], [], []
            for i in range(reps):
                pr_i = ba_prepared[i % N_PROB]
                torch.cuda.synchronize()
                t0_ = time.perf_counter(); ba.setParams(pr_i, ba_ps)
                t1_ = time.perf_counter(); ba.optimize()
                t2_ = time.perf_counter()
                check(L.uh_ba_get_results(ba._h, np_ptr(ba_out["poses"]), np_ptr(ba_out["points"]), None, np_ptr(ba_out["bad"]), np_ptr(ba_out["iters"])))
                t3_ = time.perf_counter()
                ts.append(t1_ - t0_); to.append(t2_ - t1_); tg.append(t3_ - t2_)
            return 1e3 * float(np.median(ts)), 1e3 * float(np.median(to)), 1e3 * float(np.median(tg))

        ba_phases(3)
        stage_ms["ba_set_problem_ms"], stage_ms["ba_optimize_after_set_ms"], stage_ms["ba_get_results_ms"] = ba_phases(24)
        stage_ms["ba_protocol_ms_per_keyframe"] = stage_ms["ba_set_problem_ms"] + stage_ms["ba_optimize_after_set_ms"] + stage_ms["ba_get_results_ms"]
        stage_ms["ba_ms_per_keyframe"] = timed(lambda: ba_res.optimize(), 5)   # optimize() alone on a resident problem (the kernel-side figure)
        stage_ms["ba_keyframes_per_s"] = 1e3 / stage_ms["ba_protocol_ms_per_keyframe"]   # local BAs per second through the plugin protocol (nothing else on the GPU)

        # The metric's two halves on their own, so that a reader can compose another frames-per-keyframe ratio than this bench's 4:
        # tracking only = the step WITHOUT the local BA (frames in from pinned memory, ORB, match, results out), host in / host out
        def step_tracking(fk=F):
            frames[:fk].copy_(frames_host[:fk], non_blocking=True)
            check(L.uh_orb_extract_dev(ext._h, dev_ptr(frames), W, H, W, W * H, fk, dev_ptr(kps_v), dev_ptr(desc_v), cap_kp, dev_ptr(cnt_v)))
            check(L.uh_knn_search_dev(index._h, dev_ptr(desc_v), fk * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1))
            out_host.copy_(out_dev, non_blocking=True)
            trk_stream.synchronize()

        for _ in range(3):
            step_tracking()
        stage_ms["tracking_only_step_ms"] = timed(step_tracking, 30)
        stage_ms["tracking_only_frames_per_s"] = 1e3 * F / stage_ms["tracking_only_step_ms"]

        # value at other frames-per-keyframe ratios: one local BA (fresh problem, mapper thread) beside fk frames of tracking per keyframe
        # interval — fk = 1, 2: a batch of fk frames; fk = 8: two batches of 4.  (fk = 4 is the headline step itself.)
        def step_fpk(fk):
            i = step_no[0] % N_PROB
            step_no[0] += 1
            check(L.uh_ba_solve_async(*solve_args[i]))
            for b0 in range(0, fk, F):
                step_tracking(min(F, fk - b0))
            check(L.uh_ba_wait(ba._h))
            check(L.uh_ba_get_results(*get_args))

        fpk = {}
        for fk in (1, 2, 4, 8):
            for _ in range(3):
                step_fpk(fk)
            ms_ = timed(lambda: step_fpk(fk), 20)
            fpk[str(fk)] = {"step_ms": round(ms_, 4), "frames_per_s": round(1e3 * fk / ms_, 1)}
        stage_ms["value_by_frames_per_keyframe"] = fpk

        # the frames' way in and the results' way out (pinned host memory <-> HBM), alone on the stream: 4 frames H2D + one D2H of
        # keypoints, descriptors, counts and match rows
        def io_only():
            frames.copy_(frames_host, non_blocking=True)
            out_host.copy_(out_dev, non_blocking=True)

        stage_ms["h2d_d2h_ms_per_frame"] = timed(io_only, 30) / F
        stage_ms["h2d_bytes_per_frame"] = W * H
        stage_ms["d2h_bytes_per_frame"] = out_bytes // F
        for _ in range(3):
            step_resident()
        stage_ms["kernel_only_step_ms"] = timed(step_resident, 30)
        stage_ms["kernel_only_frames_per_s"] = 1e3 * F / stage_ms["kernel_only_step_ms"]
        # the same keyframe loop from a C++ host over the C ABI (examples/track_stream.cpp: its own synthetic inputs, host buffers in and
        # out through uh_orb_extract_batch / uh_knn_search, no Python anywhere in the step) — compiled here with g++ if there is one
        try:
            import shutil, subprocess, tempfile
            if shutil.which("g++"):
                libdir = os.path.join(ROOT, "ucoslam-cv3_amd")
                exe = os.path.join(tempfile.mkdtemp(prefix="uh_cpp_"), "track_stream")
                subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "examples", "track_stream.cpp"), "-L", libdir, "-lucoslam_hip",
                                       f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                torch.cuda.synchronize()
                r_ = json.loads(subprocess.run([exe, str(args.steps), "9"], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
                stage_ms["cpp_host_step_ms"] = r_["ms_per_step"]
                stage_ms["cpp_host_frames_per_s"] = r_["frames_per_s"]
        except Exception as e_:   # (a side measurement: never fails the bench)
            stage_ms["cpp_host_step_ms"] = None
            print("cpp host stage skipped:", repr(e_), file=sys.stderr)
        # The reference tracker's real per-frame chain, ONE frame at a time, host buffers in and out of every call (examples/tracker_frame.cpp,
        # a C++ host over the C ABI): uh_orb_extract -> uh_projmatch_set_frame -> uh_projmatch_match_prev -> uh_pnp_solve -> uh_projmatch_match
        # -> uh_pnp_solve (frameextractor.cpp:430-520, frame.h:124, system.cpp:5930-6460 / :6559-6626, pnpsolver.cpp:116-409, map.cpp:651-770,
        # system.cpp:6897-6954).  Not part of the metric (ORB + match + local BA); it is what a drop-in under process() pays per frame.
        tracker = None
        try:
            if shutil.which("g++"):
                exe = os.path.join(tempfile.mkdtemp(prefix="uh_trk_"), "tracker_frame")
                subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "examples", "tracker_frame.cpp"), "-L", libdir, "-lucoslam_hip",
                                       f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                torch.cuda.synchronize()
                tracker = json.loads(subprocess.run([exe, "300", "30"], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
                stage_ms["tracker_frame_ms"] = tracker["tracker_frame_ms"]
                stage_ms["tracker_frames_per_s"] = tracker["tracker_frames_per_s"]
                # the same chain with the frame kept on the device and Frame::create_kdtree built there (uh_orb_extract_frame_dev /
                # uh_projmatch_set_frame_dev, csrc/kdbuild.hpp): identical results, no host CPU time for the tree, slower today
                trk_dev = json.loads(subprocess.run([exe, "300", "30", "dev"], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
                tracker["device_frame_route"] = {k: trk_dev[k] for k in ("tracker_frame_ms", "orb_extract_ms", "set_frame_ms", "match_prev_ms", "match_map_ms")}
                # ... and with the four tracker calls (previous-frame search, solvePnp, local-map search, solvePnp) as ONE call on a resident frame whose tree the
                # host core builds (uh_track_pose, csrc/track.hpp: list handling and look-ups on the device, six launches, one wait): identical results
                trk_f = json.loads(subprocess.run([exe, "300", "30", "fused"], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
                tracker["fused_route"] = {"tracker_frame_ms": trk_f["tracker_frame_ms"], "tracker_frames_per_s": trk_f["tracker_frames_per_s"], "orb_extract_ms": trk_f["orb_extract_ms"],
                                          "set_frame_ms": trk_f["set_frame_ms"], "track_pose_ms": trk_f["match_prev_ms"],
                                          "same_results": all(trk_f[k] == tracker[k] for k in ("keypoints", "matches_prev", "matches_map", "inliers1", "inliers2", "max_pose_err_vs_truth"))}
                stage_ms["tracker_frame_ms_fused"] = trk_f["tracker_frame_ms"]
                stage_ms["tracker_frames_per_s_fused"] = trk_f["tracker_frames_per_s"]
        except Exception as e_:
            print("tracker chain stage skipped:", repr(e_), file=sys.stderr)
        # single-frame latency, host in / host out, batch 1 (what a sequential caller sees): ORB of one pinned frame into pinned arrays, the
        # exact match of its 2000 descriptors against the 10 000-row map from / to host arrays, and the two in sequence
        single = None
        try:
            one_img = frames_host[0].numpy()
            s_k = torch.zeros(MAX_FEATURES * 28, dtype=torch.uint8).pin_memory()
            s_d = torch.zeros((MAX_FEATURES, 32), dtype=torch.uint8).pin_memory()
            s_i = torch.zeros((NQ, NN), dtype=torch.int32).pin_memory()
            s_dd = torch.zeros((NQ, NN), dtype=torch.int32).pin_memory()
            n_c = C.c_int(0)
            ext1 = ORBextractor.create(ctx)
            check(L.uh_orb_set_params(ext1._h, C.byref(fp)))
            orb1 = lambda: check(L.uh_orb_extract(ext1._h, C.c_void_p(one_img.ctypes.data), W, H, W, C.c_void_p(s_k.data_ptr()), C.c_void_p(s_d.data_ptr()), MAX_FEATURES, C.byref(n_c)))
            knn1 = lambda: check(L.uh_knn_search(index._h, C.c_void_p(s_d.data_ptr()), NQ, 32, NN, C.c_void_p(s_i.data_ptr()), C.c_void_p(s_dd.data_ptr()), 0, -1))
            for _ in range(3):
                orb1(); knn1()
            single = {"orb_extract_ms": round(timed(orb1, 50), 4), "knn_search_ms": round(timed(knn1, 50), 4)}
            single["orb_plus_match_ms"] = round(timed(lambda: (orb1(), knn1()), 50), 4)
            # the same extraction from / to PAGEABLE arrays (a cv::Mat the host never registered): the frame goes through the runtime's staging copy
            pg_img = np.array(one_img, copy=True)
            pg_k = np.zeros(MAX_FEATURES * 28, np.uint8); pg_d = np.zeros((MAX_FEATURES, 32), np.uint8)
            orb_pg = lambda: check(L.uh_orb_extract(ext1._h, C.c_void_p(pg_img.ctypes.data), W, H, W, C.c_void_p(pg_k.ctypes.data), C.c_void_p(pg_d.ctypes.data), MAX_FEATURES, C.byref(n_c)))
            for _ in range(3):
                orb_pg()
            single["orb_extract_pageable_ms"] = round(timed(orb_pg, 50), 4)
            single["note"] = "one 1241x376 frame / its 2000 x 10000 nn=10 search per call, pinned host buffers in and out, no batching"
        except Exception as e_:
            print("single-frame stage skipped:", repr(e_), file=sys.stderr)
        # the second frame size north_star asks for: the same step (4 frames, one kNN launch, one local BA) on 640x480 frames
        fr2 = torch.from_numpy(np.stack([synth.frame(640, 480, seed=77 + f, shift=(2 * f, f)) for f in range(F)])).to(dev)
        ext2 = ORBextractor.create(ctx)
        out2 = ext2.extract_batch(fr2, fp)

        fr2_host = fr2.cpu().pin_memory()

        def step_640():   # the same through-the-boundary step on the other frame size
            pr_i = ba_prepared[step_no[0] % N_PROB]
            step_no[0] += 1
            ba.solve_async(pr_i, ba_ps)
            fr2.copy_(fr2_host, non_blocking=True)
            ext2.extract_batch(fr2, fp, orb_out)
            check(L.uh_knn_search_dev(index._h, dev_ptr(desc_v), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1))
            out_host.copy_(out_dev, non_blocking=True)
            ba.wait()
            check(L.uh_ba_get_results(ba._h, np_ptr(ba_out["poses"]), np_ptr(ba_out["points"]), None, np_ptr(ba_out["bad"]), np_ptr(ba_out["iters"])))
            torch.cuda.current_stream().synchronize()

        for _ in range(3):
            step_640()
        stage_ms["step_ms_640x480"] = timed(step_640, 15)
        stage_ms["frames_per_s_640x480"] = 1e3 * F / stage_ms["step_ms_640x480"]
        # also outside the metric: a global BA (System::globalOptimization, nIters = 10) — 100 keyframes x 5000 landmarks, ~330k
        # observations — through the wide form of the same plugin object (more than 64 free keyframes)
        gba = GlobalOptimizer.create(ctx_ba)
        gba.setParams(synth.ba_problem(100, 5000, seed=1, nfixed=2), ParamSet(nIters=10))
        gba.optimize()
        stage_ms["global_ba_ms_100kf_5000pt_330kobs"] = timed(lambda: gba.optimize(), 2)
        # local BA over larger windows (the reference's window is the covisibility neighbourhood, not a constant 10: mapmanager.cpp:11085-11413):
        # optimize() of nfree + 2 keyframes x 3000 landmarks — persistent form to 16 free keyframes, launch chain (dense Schur form; fused,
        # packed-in-LDS and HBM solve) beyond
        for nfree_w in (16, 17, 24, 32, 48, 64):
            wba = GlobalOptimizer.create(ctx_ba).wantChi2(False)
            wba.setParams(synth.ba_problem(nfree_w + 2, 3000, seed=nfree_w, nfixed=2), ParamSet(nIters=5))
            wba.optimize()
            stage_ms[f"ba_optimize_ms_{nfree_w}_free_kf_3000pt"] = timed(lambda: wba.optimize(), 5)
            del wba
        del gba
        # headroom figure, NOT the metric: two independent sessions (two frame streams, two maps, two local BAs) on this one GPU —
        # the latency-bound launch chains of the two BAs interleave, which one session cannot do with itself
        # ... three and four: 94 spinning workgroups per local BA — a third does not fit the 7/8-of-the-CUs admission budget of the persistent
        # launches (ba.hip, PersistAdmission) and WAITS for one of the others to leave (it does not change form: the launch-chain form beside two
        # persistent launches is slower still, scripts/sessions_probe.py) — the figures show what that costs.  forms = uh_ba_form of each session's optimiser.
        class _Session:
            def __init__(self, k):
                self.ctx_ba, self.ctx_t = u.Context(local_rank, private=True), u.Context(local_rank, private=True)
                self.ba = GlobalOptimizer.create(self.ctx_ba).wantChi2(False)
                self.ba.setParams(synth.ba_problem(BA_K, BA_P, seed=rank + 100 * k), ParamSet(nIters=5))
                self.ext = ORBextractor.create(self.ctx_t)
                self.idx = Index(self.ctx_t).build(map_desc)
                self.out = self.ext.extract_batch(frames, fp)
                self.knn_idx, self.knn_dist = torch.empty_like(knn_idx), torch.empty_like(knn_dist)

        import types

        sess = [types.SimpleNamespace(ba=ba_res, ext=ext, idx=index, out=orb_out, knn_idx=knn_idx, knn_dist=knn_dist)]   # the bench's own objects as the first session
        sess += [_Session(k) for k in (1, 2, 3)]

        def step_sessions(n):   # (resident form, like kernel_only_*)
            for s_ in sess[:n]: s_.ba.optimize_async()
            for s_ in sess[:n]: s_.ext.extract_batch(frames, fp, s_.out)
            for s_ in sess[:n]: check(L.uh_knn_search_dev(s_.idx._h, dev_ptr(s_.out[1]), F * NQ, NN, dev_ptr(s_.knn_idx), dev_ptr(s_.knn_dist), 0, -1))
            for s_ in sess[:n]: s_.ba.wait()

        sess_ms = {1: stage_ms["kernel_only_step_ms"]}
        for n_s in (2, 3, 4):
            for _ in range(3):
                step_sessions(n_s)
            sess_ms[n_s] = timed(lambda: step_sessions(n_s), 15)
        stage_ms["two_sessions_step_ms"] = sess_ms[2]
        stage_ms["two_sessions_frames_per_s"] = 1e3 * 2 * F / sess_ms[2]
        stage_ms["sessions_on_one_gpu"] = {
            "frames_per_s": {str(n_s): round(1e3 * n_s * F / sess_ms[n_s], 1) for n_s in (1, 2, 3, 4)},
            "step_ms": {str(n_s): round(sess_ms[n_s], 4) for n_s in (1, 2, 3, 4)},
            "ba_forms": [s_.ba.form() for s_ in sess],
            "note": "resident form (frames in HBM, one problem per session re-optimised); persistent local-BA launches are admitted up to 7/8 of the CUs: two run side by side, "
                    "a third and a fourth wait their turn (412 registers per lane leave no room for a second landmark group per workgroup: DESIGN.md section 4.3)"}
        del sess, step_sessions
        import gc
        gc.collect()   # (the three extra sessions' contexts, optimisers with their worker threads and buffers go away before the remaining stages are timed)
        stage_ms["orb_ms_per_frame_640x480"] = timed(lambda: ext2.extract_batch(fr2, fp, out2), 20) / F
        # not part of the metric's step (ORB + match + local BA): the per-frame pose-only solve (PnPSolver::solvePnp, 600 matches)
        from ucoslam_cv3_amd.pnp import PnPSolver

        pnp_pr = synth.pnp_problem(600, seed=3)
        pnp = PnPSolver(ctx)
        pd = {k: torch.from_numpy(np.ascontiguousarray(pnp_pr[k])).to(dev) for k in ("pose", "intr", "p3d", "kp", "invsig", "weight")}
        pwork = torch.empty(600 * 32, dtype=torch.uint8, device=dev)
        pout = (torch.empty(16, dtype=torch.float32, device=dev), torch.empty(600, dtype=torch.uint8, device=dev),
                torch.empty(5, dtype=torch.int32, device=dev), torch.empty(7, dtype=torch.float64, device=dev))
        stage_ms["pnp_ms_per_solve_600_matches"] = timed(lambda: check(L.uh_pnp_solve_dev(
            pnp._h, dev_ptr(pd["pose"]), dev_ptr(pd["intr"]), 600, dev_ptr(pd["p3d"]), dev_ptr(pd["kp"]), dev_ptr(pd["invsig"]), dev_ptr(pd["weight"]),
            dev_ptr(pwork), dev_ptr(pout[0]), dev_ptr(pout[1]), dev_ptr(pout[2]), dev_ptr(pout[3]))), 20)
        # also outside the metric's step: what FrameMatcher_Flann really runs per frame pair — xflann's hierarchical k-means index (k = 32) built on
        # the train descriptors and searched with 16 checks for the frame's 2000 queries; the CPU leg beside it is
        # cpu_baseline.match_hkmeans32_checks16_build_plus_search_ms_1_thread.  Round 6: the build runs level by level — shuffles (replayed from
        # libstdc++'s recorded swaps), centres and the byte layout on the host, the distances of a whole level in one device launch.
        try:
            km_train, _ = synth.match_set(1, NT, seed=7)
            km_index = Index(ctx)
            for _ in range(3):
                km_index.build_kmeans(km_train, 32, 0)
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            for _ in range(10):
                km_index.build_kmeans(km_train, 32, 0)
            torch.cuda.synchronize()
            stage_ms["hkmeans32_build_ms_10000_rows"] = 1e3 * (time.perf_counter() - t_b) / 10
            km_pair = Index(ctx)   # one frame pair of the matcher: the train FRAME's 2000 descriptors, build + search
            km_tr2 = np.ascontiguousarray(km_train[:MAX_FEATURES])
            km_q0 = orb_out[1][0]
            for _ in range(3):
                km_pair.build_kmeans(km_tr2, 32, 0); km_pair.search_kmeans(km_q0, NN, 16, sorted=False)
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            for _ in range(10):
                km_pair.build_kmeans(km_tr2, 32, 0); km_pair.search_kmeans(km_q0, NN, 16, sorted=False)
            torch.cuda.synchronize()
            stage_ms["hkmeans32_frame_pair_build_plus_search_ms_2000x2000"] = 1e3 * (time.perf_counter() - t_b) / 10
            km_q = orb_out[1][0]
            km_index.search_kmeans(km_q, NN, 16, sorted=False)
            stage_ms["hkmeans32_search_ms_2000q_checks16"] = timed(lambda: km_index.search_kmeans(km_q, NN, 16, sorted=False), 30)
        except Exception as e_:
            print("hkmeans stage skipped:", repr(e_), file=sys.stderr)
        # also outside the metric's step: the projection matcher (Map::matchFrameToMapPoints), 2000 keypoints x 3000 candidate
        # map points, host buffers in and out as the reference's call site has them (includes one H2D and one D2H)
        from ucoslam_cv3_amd.projmatch import ProjectionMatcher

        pfr, pmp, ppose = synth.proj_problem(2000, 3000, 0)
        pmatch = ProjectionMatcher(ctx)
        pmatch.setFrame(pfr["und_kpts"], pfr["desc"], pfr["scale_factors"], pfr["fx"], pfr["fy"], pfr["cx"], pfr["cy"], pfr["min_xy"], pfr["max_xy"])
        stage_ms["projmatch_ms_per_call_2000kp_3000pts"] = timed(lambda: pmatch.matchFrameToMapPoints(
            ppose, pmp["ids"], pmp["pos3d"], pmp["normal"], pmp["min_dist"], pmp["max_dist"], pmp["desc"], 100.0, 15.0), 20)
        # ... and the tracker's search against the previous frame (system.cpp:5930-6460), called with (1.5 * maxDescDistance, projDistThr)
        stage_ms["projmatch_prev_ms_per_call_2000kp_3000pts"] = timed(lambda: pmatch.matchFrameToPrevFrame(
            ppose, pmp["ids"], pmp["pos3d"], pmp["octave"], pmp["desc"], 75.0, 15.0), 20)
        ba_iters = [int(v) for v in ba_out["iters"]]   # (of the last step's problem)
        if not args.no_roofline:
            for c in (ctx, ctx_ba):
                c.prof_enable(True)
                c.prof_reset()
            reps = 8
            it_sum, e_sum = 0, 0
            for _ in range(reps):
                e_sum += ba_problems[step_no[0] % N_PROB]["E"]
                step()
                it_sum += int(ba_out["iters"].sum())
            torch.cuda.synchronize()
            rep = dict(ctx.prof_report())
            rep.update(ctx_ba.prof_report())
            for c in (ctx, ctx_ba):
                c.prof_enable(False)
            import re as _re

            short = {}
            for kname, v in rep.items():          # "(anonymous namespace)::ba_persist_kernel<8>" -> "ba_persist_kernel"
                n = _re.sub(r"<.*>", "", kname.split("::")[-1]).strip("()")
                c0, t0_ = short.get(n, (0, 0.0))
                short[n] = (c0 + v[0], t0_ + v[1])
            # ---- algorithmic bytes per unit x units per launch / launch time / 8 TB/s
            b8d = survey_8d_bytes(MAX_FEATURES, e_sum // reps)     # (E = the profiled problems' average number of observations)
            lm_iters = it_sum / reps                               # LM iterations one local BA executes (5 + 10 on these problems)
            unit_bytes = {"orb": F * b8d["orb_per_frame"], "match": F * b8d["match_per_frame"], "ba": lm_iters * b8d["ba_per_lm_iteration"]}
            unit_ms = {"orb": sum(v[1] for k, v in short.items() if k in ORB_KERNELS) / reps,
                       "match": sum(v[1] for k, v in short.items() if k in MATCH_KERNELS) / reps,
                       "ba": sum(v[1] for k, v in short.items() if k.startswith("ba_")) / reps}
            dom = max(short.items(), key=lambda kv: kv[1][1])
            name, (calls, tot_ms) = dom
            avg_ms = tot_ms / max(calls, 1)
            if name.startswith("ba_"):
                # the persistent kernel is ONE launch per local BA = lm_iters LM iterations; legacy BA kernels run once per trial
                launches_per_ba = calls / reps
                alg = unit_bytes["ba"] / launches_per_ba
                units = f"{lm_iters:g} LM iterations per launch x {b8d['ba_per_lm_iteration']} B (E*32 + 2P*24 + 2K*56 + K^2*288)" if launches_per_ba <= 1.01 else \
                    f"{b8d['ba_per_lm_iteration']} B per LM iteration / {launches_per_ba / lm_iters:.2f} launches of this kernel per iteration"
            elif name in MATCH_KERNELS:
                alg, units = unit_bytes["match"], f"{F} frames per launch x {b8d['match_per_frame']} B ((NQ+NT)*32 + NQ*k*8, k=10)"
            else:   # one ORB kernel: the extractor's §8(d) bytes are stated for the whole extractor; this kernel's share by time
                alg = unit_bytes["orb"] * (tot_ms / reps) / max(unit_ms["orb"], 1e-9)
                units = f"{F} frames x {b8d['orb_per_frame']} B (whole extractor), this kernel's share by time"
            achieved = alg / (avg_ms * 1e-3) / 1e9
            traffic = impl = None
            trials_est = int(round(lm_iters)) + 2
            if name == "ba_persist_kernel":
                impl, g_wg = persistent_ba_exchange_bytes(ba_pr["P"], trials_est)
            # HBM bytes per launch from the committed PMC passes (profiles/, scripts/collect_profiles*.sh; separate FETCH_SIZE / WRITE_SIZE passes of
            # `bench.py --quick`): NOT measured in this run.  Corrected as the guide prescribes for gfx950: FETCH_SIZE counts wide reads once -> x2.
            pmc_all, pmc_src = {}, None
            try:
                pmc_file = next(f_ for f_ in (os.path.join(ROOT, "profiles", n_) for n_ in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")) if os.path.exists(f_))
                pmc_all = json.load(open(pmc_file))["kernels"]
                pmc_src = "profiles/" + os.path.basename(pmc_file) + " (fetch_bytes_x2 + write_bytes per launch; separate rocprofv3 --pmc passes of `bench.py --quick`, not measured in this run)"
                pmc = pmc_all.get(name)
                if pmc:
                    traffic = pmc.get("fetch_bytes_x2", 2 * pmc["fetch_bytes"]) + pmc["write_bytes"]
            except Exception:
                traffic = None
            step_bytes = unit_bytes["orb"] + unit_bytes["match"] + unit_bytes["ba"]
            roofline = {
                "bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": pmc_src,
                "traffic_ratio": round(traffic / alg, 2) if traffic else None,
                "avg_launch_us": round(avg_ms * 1e3, 3), "algorithmic_bytes_per_launch": int(alg), "units_per_launch": units,
                "impl_bytes_per_launch": impl,
                "share_of_step_gpu_time": round(tot_ms / sum(v[1] for v in short.values()), 4),
                # the whole step by the same §8(d) figures: (F*B_orb + F*B_match + iters*B_ba) / step time / peak
                "step_bytes": int(step_bytes), "step_gbps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                "step_frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "stage_gbps": {k: round(unit_bytes[k] / (unit_ms[k] * 1e-3) / 1e9, 2) for k in unit_bytes if unit_ms[k] > 0},
                "stage_gpu_ms_per_step": {k: round(v, 4) for k, v in unit_ms.items()},
                "kernels_us": {k: round(1e3 * v[1] / max(v[0], 1), 2) for k, v in sorted(short.items())},
            }
            # the other stages with THEIR bounds stated: the exact matcher is integer-ALU work (XOR + popcount + heap test per (query, train)
            # pair), the extractor's kernels stream the pyramid
            others = {}
            knn_name = max((k for k in short if k in MATCH_KERNELS), key=lambda k: short[k][1], default=None)
            if knn_name:
                kc, kt = short[knn_name]
                k_ms = kt / max(kc, 1)
                pairs = F * NQ * NT
                ops_per_pair = 16                                       # 4 x (v_xor + v_bcnt accumulate) on 64-bit halves + compare / ballot / index bookkeeping
                valu_peak = 256 * 4 * 32 * 2.4e9                        # lane-ops/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)
                kp_ = pmc_all.get(knn_name)
                others[knn_name] = {
                    "avg_launch_us": round(k_ms * 1e3, 2),
                    "valu": {"bound": "valu", "achieved": round(pairs * ops_per_pair / (k_ms * 1e-3) / 1e12, 3), "peak": round(valu_peak / 1e12, 2), "unit": "T lane-ops/s",
                             "frac": round(pairs * ops_per_pair / (k_ms * 1e-3) / valu_peak, 4), "units_per_launch": f"{pairs} (query, train) pairs x ~{ops_per_pair} VALU lane-ops"},
                    "hbm": {"bound": "hbm", "achieved": round(unit_bytes["match"] / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(unit_bytes["match"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": int(unit_bytes["match"]),
                            "traffic": (kp_.get("fetch_bytes_x2", 2 * kp_["fetch_bytes"]) + kp_["write_bytes"]) if kp_ else None},
                }
            if unit_ms["orb"] > 0:
                orb_traffic = sum((pmc_all[k].get("fetch_bytes_x2", 2 * pmc_all[k]["fetch_bytes"]) + pmc_all[k]["write_bytes"]) * (short[k][0] / reps) for k in short if k in ORB_KERNELS and k in pmc_all)
                others["orb_kernels"] = {"bound": "hbm", "achieved": round(unit_bytes["orb"] / (unit_ms["orb"] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(unit_bytes["orb"] / (unit_ms["orb"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "gpu_ms_per_step": round(unit_ms["orb"], 4),
                                         "algorithmic_bytes_per_step": int(unit_bytes["orb"]), "traffic_per_step": int(orb_traffic) if orb_traffic else None,
                                         "note": f"{F} frames per step; launch-latency bound at this size (8-9 dependent launches of a few microseconds each)"}
            # the second frame size north_star names: the same step on 640 x 480 frames (SURVEY 8(d) bytes of that size over the measured times)
            n640 = int(out2[2].min().item()) if hasattr(out2[2], "min") else MAX_FEATURES
            b640 = survey_8d_bytes(n640, e_sum // reps, w=640, h=480)
            step640_bytes = F * b640["orb_per_frame"] + F * b640["match_per_frame"] + unit_bytes["ba"]
            others["frame_640x480"] = {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "keypoints_per_frame": n640,
                "orb_algorithmic_bytes_per_frame": int(b640["orb_per_frame"]), "orb_ms_per_frame": round(stage_ms["orb_ms_per_frame_640x480"], 4),
                "orb_achieved": round(b640["orb_per_frame"] / (stage_ms["orb_ms_per_frame_640x480"] * 1e-3) / 1e9, 2),
                "orb_frac": round(b640["orb_per_frame"] / (stage_ms["orb_ms_per_frame_640x480"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "step_ms": round(stage_ms["step_ms_640x480"], 4), "frames_per_s": round(stage_ms["frames_per_s_640x480"], 1),
                "step_bytes": int(step640_bytes), "step_achieved": round(step640_bytes / (stage_ms["step_ms_640x480"] * 1e-3) / 1e9, 2),
                "step_frac": round(step640_bytes / (stage_ms["step_ms_640x480"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "note": "the step is the local-BA protocol at this size too (the BA does not depend on the frame size); the extractor alone is launch-latency bound"}
            roofline["others"] = others
        # the metric's serial definition (1 / (t_ORB + t_match + t_BA amortised)) beside the overlapped headline
        t_serial = stage_ms["orb_ms_per_frame"] + stage_ms["match_ms_per_frame"] + stage_ms["h2d_d2h_ms_per_frame"] + stage_ms["ba_protocol_ms_per_keyframe"] / F
        stage_ms["serial_frames_per_s"] = 1e3 / t_serial

    # ---- sharded frame stream (N > 1): levels + train tiles sharded, each rank holding ONLY its tile, ONE fused all-gather per frame
    def run_sharded():
        sharded = None
        if dist is not None and (world > 1 or os.environ.get("UH_BENCH_SHARDED")):   # (the env switch exercises the stage with one rank)
            from ucoslam_cv3_amd import parallel

            b = parallel.shard_bounds(NT, world)
            map0_np, _ = synth.match_set(1, NT, seed=50)                       # the SAME map on every rank, each keeps its tile
            tile = Index(ctx).build(torch.from_numpy(map0_np[b[rank]:b[rank + 1]].copy()).to(dev)).set_row_offset(b[rank])
            ext_s = ORBextractor.create(ctx)
            # the stream below Python (uh_fstream_*, csrc/fstream.hip): producers write into the message, ONE RCCL all-gather per frame through
            # the library's own communicator on the tracking stream, the replay reads the gathered lists in place, no host synchronisation
            # accept-list capacity per (query, tile): a tile scanned from an empty heap accepts k (1 + ln(rows / k)) rows on average (a record
            # process, spread ~ its square root): mean + 4 sigma, rounded up to 16 — a list that overflows would invalidate the frame
            rows_t = max(b[rank + 1] - b[rank], 1)
            exp_acc = NN * (1.0 + np.log(max(rows_t / NN, 1.0)))
            cand_cap = int(-(-(exp_acc + 4.0 * np.sqrt(exp_acc)) // 16) * 16)
            stream = parallel.ShardedFrameStreamDev(ctx, ext_s, fp, tile, NN, MAX_FEATURES, cand_cap=cand_cap, rank=rank, world=world, device=dev)
            if world > 1:
                stream.init_comm()
            sframes = [torch.from_numpy(synth.frame(W, H, seed=9000 + f, shift=(2 * f, f))).to(dev) for f in range(4)]
            for i in range(6):
                stream.step(sframes[i % 4])
            sync_all()
            n_s = 40
            t0 = time.perf_counter()
            ovf = 0
            for i in range(n_s):
                r = stream.step(sframes[i % 4])
            torch.cuda.synchronize()
            ts = time.perf_counter() - t0
            ovf = int(r["overflow"])
            tt = torch.tensor([ts], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sharded = {"rccl_ranks": stream.comm_ranks(),   # ncclCommCount of the library's own communicator: what the collective really spans
                       "sharded_frame_ms": round(1e3 * float(tt.item()) / n_s, 4), "sharded_frames_per_s": round(n_s / float(tt.item()), 2),
                       "collectives_per_frame": 1, "message_bytes_per_rank": stream.message_bytes, "cand_cap": cand_cap, "train_rows_per_rank": b[rank + 1] - b[rank],
                       "levels_of_rank0": list(parallel.level_ranges(W, H, NLEVELS, SCALE, world)[0]), "overflow": ovf}
            if rank == 0:   # the same frame stream un-sharded on one GPU: one frame per launch sequence + full search (latency form)
                full = Index(ctx).build(torch.from_numpy(map0_np).to(dev))
                one = sframes[0][None]
                o1 = ext_s.extract_batch(one, fp)

                def single():
                    k, d_, c = ext_s.extract_batch(one, fp, o1)
                    full.search(d_[0], NN)

                for _ in range(3):
                    single()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_s):
                    single()
                torch.cuda.synchronize()
                sharded["single_gpu_frame_ms"] = round(1e3 * (time.perf_counter() - t0) / n_s, 4)
        return sharded

    # ---- CPU baseline (rank 0, N=1 only): bounded sample of the same workload on the host cores, in the shapes SURVEY §8(d) asks for
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib

        O = oracle_lib.load_oracle()
        ncores = os.cpu_count() or 1

        import threading

        def threaded_rate(nthreads, per_thread, fn):   # calls per second with `nthreads` host threads (ctypes releases the GIL inside the call)
            def work():
                for i in range(per_thread):
                    fn(i)

            ts = [threading.Thread(target=work) for _ in range(nthreads)]
            t = time.perf_counter()
            for x in ts:
                x.start()
            for x in ts:
                x.join()
            return nthreads * per_thread / (time.perf_counter() - t)

        oracle_lib.orb_extract(O, frames_np[0], MAX_FEATURES, NLEVELS, SCALE)   # (sets the ctypes signature once, before threads call it)
        orb_fn = lambda i: oracle_lib.orb_extract(O, frames_np[i % F], MAX_FEATURES, NLEVELS, SCALE)
        orb_1 = threaded_rate(1, 8, orb_fn)
        orb_2 = threaded_rate(2, 6, orb_fn)
        orb_all = threaded_rate(ncores, 3, orb_fn)
        q = orb_out[1][0].cpu().numpy()
        xf = oracle_lib.load_ref("xflann")
        P = oracle_lib.P

        def match_fn(i):
            ii = np.empty((NQ, NN), np.int32)
            dd = np.empty((NQ, NN), np.int32)
            if xf is not None:
                xf.xflann_ref_linear_search(P(map_desc_np), NT, P(q), NQ, NN, 0, 1, P(ii), P(dd))
            else:
                oracle_lib.knn_search(O, map_desc_np, q, NN, 0)

        # xflann's own KnnSearchParams(threads > 1) cannot be used: Index::parallel_search (index.cpp:109-126) captures its loop variables
        # by reference in the thread lambdas and corrupts the heap (reproduced here with threads = 2); "all cores" = that many
        # independent single-threaded searches side by side
        m_1 = 1e3 / threaded_rate(1, 8, match_fn)
        m_all = 1e3 / threaded_rate(ncores, 3, match_fn)
        # the matcher the reference's FrameMatcher_Flann actually runs (framematcher.cpp:213,239): a hierarchical k-means index built per
        # train frame (k = 32, maxIters = 0) and searched with nn = 10, maxChecks = 16 — build + search, real xflann
        hk_ms = None
        if xf is not None:
            t_ = time.perf_counter()
            for _ in range(3):
                oracle_lib.ref_hkmeans_search(xf, map_desc_np, q, NN, 32, 0, 16, 0)
            hk_ms = 1e3 * (time.perf_counter() - t_) / 3
        g2o = oracle_lib.load_ref("g2o")
        n_ba = 4
        t = time.perf_counter()
        for i in range(n_ba):
            if g2o is not None:
                oracle_lib.ba_optimize_ref(g2o, ba_pr, 5)
            else:
                oracle_lib.ba_optimize(O, ba_pr, 5)
        ba_ms = 1e3 * (time.perf_counter() - t) / n_ba
        # real-g2o legs for the larger local-BA windows the bench times on the GPU (stages.ba_optimize_ms_<n>_free_kf_3000pt): one optimisation each
        ba_windows_cpu = {}
        if g2o is not None:
            for nfree_w in (16, 32, 64):
                pr_w = synth.ba_problem(nfree_w + 2, 3000, seed=nfree_w, nfixed=2)
                t_ = time.perf_counter()
                oracle_lib.ba_optimize_ref(g2o, pr_w, 5)
                ba_windows_cpu[str(nfree_w)] = round(1e3 * (time.perf_counter() - t_), 1)
        # the reference's own arrangement: extractor with nthreads = 2 (ucoslamtypes.cpp:40), matcher 1 thread, g2o 1 thread (config.h:7)
        t_ref = 1e3 / orb_2 + m_1 + ba_ms / F
        t_all = 1e3 / orb_all + m_all + ba_ms / F
        cpu = {
            "value": round(1e3 / t_ref, 4), "unit": "frames/s", "cores": 2, "host_cores": ncores,
            "kind": "port",
            "value_all_cores": round(1e3 / t_all, 4), "value_one_core": round(1e3 / (1e3 / orb_1 + m_1 + ba_ms / F), 4),
            "orb_frames_per_s": {"1_thread": round(orb_1, 2), "2_threads": round(orb_2, 2), f"{ncores}_threads": round(orb_all, 2)},
            "match_ms_2000x10000_nn10": {"1_thread": round(m_1, 2), f"{ncores}_threads_throughput": round(m_all, 2)},
            "match_hkmeans32_checks16_build_plus_search_ms_1_thread": round(hk_ms, 2) if hk_ms is not None else None,
            "ba_ms_per_keyframe_1_thread": round(ba_ms, 2),
            "ba_ms_by_free_keyframes_3000pt_real_g2o_1_thread": ba_windows_cpu,
            "sample": (f"ORB = this repo's oracle port (OpenCV absent: the reference extractor cannot be built), 8/12/{3 * ncores} frames at 1/2/{ncores} threads "
                       f"(whole frames per thread: an upper bound for the reference's level-parallel nthreads); matcher = "
                       f"{'real xflann Linear (oracle/_ref)' if xf is not None else 'oracle port'}, 8 searches on 1 thread and {3 * ncores} on {ncores} threads (independent searches: the reference's threads > 1 path crashes), 2000x10000 nn=10; "
                       f"BA = {'real g2o (oracle/_ref)' if g2o is not None else 'oracle port'}, {n_ba} local BAs, single-threaded like g2o; "
                       f"value = 1/(t_ORB(2 threads) + t_match(1 thread) + t_BA/{F}), the reference's default threading"),
        }

    # CPU leg of the tracker's per-frame chain (rank 0, N=1): the same stages on the host cores, one thread — oracle ORB ("port"), this repo's
    # restated projection matchers over the restated picoflann ("port"; both pinned to the real picoflann / real g2o where DESIGN.md says so),
    # PnP through the REAL g2o (oracle/_ref, "reference")
    if rank == 0 and cpu is not None:
        try:
            pfr_, pmp_, ppose_ = synth.proj_problem(2000, 3000, 0)
            prev_ = {k_: v_[:800] for k_, v_ in pmp_.items()}
            t_ = time.perf_counter(); oracle_lib.proj_match_prev(O, pfr_, prev_, ppose_, 75.0, 15.0); t_prev = 1e3 * (time.perf_counter() - t_)
            t_ = time.perf_counter(); oracle_lib.proj_match(O, pfr_, pmp_, ppose_, 100.0, 15.0); t_map = 1e3 * (time.perf_counter() - t_)
            pn1, pn2 = synth.pnp_problem(800, seed=3), synth.pnp_problem(1300, seed=4)
            solver = (lambda pr_: oracle_lib.pnp_solve_ref(g2o, pr_)) if g2o is not None else (lambda pr_: oracle_lib.pnp_solve(O, pr_))
            t_ = time.perf_counter(); solver(pn1); t_p1 = 1e3 * (time.perf_counter() - t_)
            t_ = time.perf_counter(); solver(pn2); t_p2 = 1e3 * (time.perf_counter() - t_)
            t_orb = 1e3 / orb_1
            cpu["tracker_chain"] = {
                "frame_ms": round(t_orb + t_prev + t_p1 + t_map + t_p2, 3), "frames_per_s": round(1e3 / (t_orb + t_prev + t_p1 + t_map + t_p2), 2), "cores": 1,
                "orb_ms": round(t_orb, 3), "match_prev_800_ms_incl_kdtree": round(t_prev, 3), "pnp_800_ms": round(t_p1, 3), "match_map_3000_ms_incl_kdtree": round(t_map, 3),
                "pnp_1300_ms": round(t_p2, 3),
                "kind": {"orb": "port", "projection matchers": "port", "pnp": "reference" if g2o is not None else "port"},
                "sample": "one call per stage on synthetic inputs of the tracker_frame example's sizes (2000 keypoints, 800 previous-frame items, 3000 map points, 800 / 1300 matches)"}
        except Exception as e_:
            print("cpu tracker leg skipped:", repr(e_), file=sys.stderr)

    if rank == 0:
        line = {
            "metric": "tracking frames/sec (ORB extract + match + local BA), 1241x376 mono",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_min": round(ms_min, 4), "ms_per_step_max": round(ms_max, 4), "reps": len(reps_s),
            "timing": "median over reps of (K steps between barrier + synchronize) / K; value = frames / that",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (ORB, Hamming) + f64 (BA)", "data": "synthetic",
            "config": {"workload": "orb1241x376_2000f_8lv + hamming_knn_2000x10000_nn10 + local_ba_10kf_3000pt",
                       "frames_per_step": F, "frames_per_keyframe": F,
                       "boundary": "host in / host out: frames from pinned host memory, keypoints + descriptors + match rows back to it; local BA = a fresh problem per keyframe through setParams / optimize / getResults on flattened host arrays",
                       "parallelism": f"frame-streams x{world} (replicas, no data-path collective; value = overlapped tracking + local-BA streams per GPU, stages.serial_frames_per_s = the serial 8(d) definition"
                       + (f"; stages.sharded_* = ONE stream over {world} GPUs, levels + train tiles sharded, one RCCL all-gather per frame)" if world > 1 else ")")},
            "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stage_ms.items()}, "keypoints_per_frame": int(counts.min()), "full_budget": full_frames,
            "ba_lm_iterations": ba_iters if rank == 0 else None,
            "tracker_chain": tracker, "single_frame_latency": single,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 2)
            line["gpu_serial_over_cpu"] = round(stage_ms["serial_frames_per_s"] / cpu["value"], 2)

    def emit_and_exit(code=0):
        # RCCL writes its version banner through C stdio, which a pipe buffers until exit: drain it first so that the JSON
        # line is the LAST line on stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        if rank == 0:
            print(json.dumps(line), flush=True)
        os._exit(code)

    # The sharded stage is a SIDE measurement whose collective has never run on more than one GPU anywhere: it runs last, under a
    # watchdog, so that neither an exception nor a hang in it can cost the headline line (measured above, assembled already).
    if dist is not None and (world > 1 or os.environ.get("UH_BENCH_SHARDED")):
        import threading

        dog = threading.Timer(120.0, lambda: (line["stages"].update({"sharded_error": "timed out after 120 s"}) if rank == 0 else None, emit_and_exit(0)))
        dog.daemon = True
        dog.start()
        try:
            sharded = run_sharded()
            if sharded and world > 1:
                assert sharded["rccl_ranks"] == world, f"the sharded stream's communicator spans {sharded['rccl_ranks']} ranks, not {world}"
            if rank == 0 and sharded:
                line["stages"].update({k: v for k, v in sharded.items()})
                # the two multi-GPU lines side by side: replicas (the metric: N independent streams) and ONE stream sharded over the N GPUs
                line["multi_gpu"] = {"replica_frames_per_s": line["value"], "sharded_one_stream_frames_per_s": sharded["sharded_frames_per_s"],
                                     "rccl_ranks": sharded["rccl_ranks"], "n_gpus": world,
                                     "note": "value = replicas (weak scaling, no data-path collective); the sharded line is ONE frame stream with levels + train tiles + fbow slices over the ranks and one RCCL all-gather per frame — by this repository's own accounting it cannot beat a single GPU at this frame size (DESIGN.md section 6)"}
        except Exception as e_:
            if rank == 0:
                line["stages"]["sharded_error"] = repr(e_)[:300]
            print("sharded stage failed:", repr(e_), file=sys.stderr)
            dog.cancel()
            emit_and_exit(0)   # (the other ranks may be inside a collective: no barrier, leave at once)
        dog.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which a pipe buffers until exit: drain it first so that the JSON
        # line is the LAST line on stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    else:
        line = None


if __name__ == "__main__":
    main()
