#!/usr/bin/env python
"""bench.py — tracking frames/sec of the MI355X hot path (ORB extract + Hamming match + local BA), 1241x376 mono.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run, one
rank per GPU.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1..3] combined = the metric's "ORB extract + match + local BA"):
  one step = `frames_per_step` (4) synthetic 1241x376 frames already resident in HBM:
     ORB extraction of the 4 frames in one batched launch set (8 levels, scale 1.2, 2000 features)
     4 x brute-force Hamming kNN, 2000 query descriptors (the frame's own ORB output) vs a 10 000-descriptor map,
         nn=10 unsorted — the FrameMatcher_Flann call shape (framematcher.cpp:213,239)
     1 x local BA, 10 keyframes x 3000 landmarks (~26k observations), nIters=5 (+10), fp64 — one keyframe per 4 frames
  value = frames / second over all ranks.  Multi-GPU: frame streams are independent, so every rank runs the same per-GPU
  workload on its own frames ("weak" scaling, no data-path collective); only the timing reduction crosses ranks.
Extra objects: "roofline" (dominant kernel, HIP events on the launch stream) and "cpu_baseline" (rank 0, N=1).
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


def level_sizes():
    inv, s = [], 1.0
    sc = np.float32(1.0)
    out = []
    for l in range(NLEVELS):
        out.append((int(np.rint(np.float32(W) * (np.float32(1.0) / sc))), int(np.rint(np.float32(H) * (np.float32(1.0) / sc)))))
        sc = np.float32(sc * np.float32(SCALE))
    return out


def algorithmic_bytes(frames_per_step, ba_E):
    """Algorithmic (compulsory) bytes PER LAUNCH of each kernel for this workload; formulas are stated in DESIGN.md §5."""
    lv = level_sizes()
    px = [w * h for w, h in lv]
    F = frames_per_step
    sum_px = sum(px)
    b = {
        "blur7_kernel": F * 2 * px[0],                                  # read input, write level 0
        "fast_score_kernel": F * 2 * sum_px,                            # read every level once, write its strength map
        "cell_nms_kernel": F * (sum(max(w - 38, 0) * max(h - 38, 0) for w, h in lv)),   # read cell interiors (candidates are ~KBs)
        "select_kernel": F * MAX_FEATURES * 4 * 4,                      # candidate words in, selected words out (order of magnitude)
        "describe_kernel": F * MAX_FEATURES * (961 + 60),               # 31x31 patch per keypoint + 28 B keypoint + 32 B descriptor
        "knn_search_kernel": (F * NQ + NT) * 32 + F * NQ * NN * 8,      # SURVEY §8(d) formula with k=10, F frames per launch
        # BA, per launch (E*32 obs + points/poses; per-kernel split in DESIGN.md)
        "ba_lin_kernel": ba_E * (32 + 18 * 8) + BA_P * (24 + 96),       # once per pass: obs in, Hpl / Hll / bl out
        # schur: per landmark pair blocks read Hpl (18 doubles per edge) and Hll/bl (12 per point); the camera workgroups
        # re-read the observations (32 B) for Hpp/bp; + the previous trial's partial sums for the decision (~3 KB)
        "ba_schur_kernel": ba_E * (18 * 8 + 32) + BA_P * 96 + 3072,
        # backsub (with the reduced-system solve inside): pair partials 36 x 12 chunks x 42 doubles + camera partials
        # 8 x 8 x 27 doubles, then per edge Hpl in (144 B), obs (32 B), errors/chi2 out (24 B) and the trial linearisation
        # out (144 B), per point Hll/bl in (96 B), point in/out (48 B), Hll/bl out (96 B)
        "ba_backsub_kernel": 36 * 12 * 42 * 8 + 8 * 8 * 27 * 8 + ba_E * (144 + 32 + 24 + 144) + BA_P * (96 + 48 + 96),
        "ba_solve_kernel": 36 * 12 * 42 * 8 + 8 * 8 * 27 * 8 + 4 * 48 * 8 + 10 * 19 * 8,   # only for n > 120 (not this workload)
        "ba_decide_kernel": 3072,
    }
    # resize: the driver launches it once per level l>=1: read level l-1, write level l (average per launch)
    b["resize_cubic_kernel"] = F * (sum(px[:-1]) + sum(px[1:])) / (NLEVELS - 1)
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
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
    ctx = u.Context(local_rank, torch.cuda.current_stream().cuda_stream)

    # ---- synthetic inputs, resident in HBM before the timed region
    frames_np = np.stack([synth.frame(W, H, seed=1000 * rank + f, shift=(2 * f, f)) for f in range(F)])
    frames = torch.from_numpy(frames_np).to(dev)
    map_desc_np, _ = synth.match_set(1, NT, seed=50 + rank)
    map_desc = torch.from_numpy(map_desc_np).to(dev)
    ba_pr = synth.ba_problem(BA_K, BA_P, seed=rank)

    ext = ORBextractor.create(ctx)
    fp = FeatParams(MAX_FEATURES, NLEVELS, SCALE)
    orb_out = ext.extract_batch(frames, fp)
    index = Index(ctx).build(map_desc)
    # the matcher runs beside the local BA: four queries per wave = a quarter of the L1/L2 streaming and of the resident waves; the
    # search alone takes longer (250 instead of 163 us for the 4 x 2000 queries), the step is shorter
    index.set_queries_per_wave(4)
    # the reference runs local BA on its mapper thread, concurrently with tracking (mapmanager.cpp:1550);
    # here BA gets its own HIP stream so that its latency-bound launch chain overlaps the tracking stream's kernels
    ctx_ba = u.Context(local_rank, private=True)
    ba = GlobalOptimizer.create(ctx_ba)
    ba.setParams(ba_pr, ParamSet(nIters=5))
    knn_idx = torch.empty((F, NQ, NN), dtype=torch.int32, device=dev)
    knn_dist = torch.empty((F, NQ, NN), dtype=torch.int32, device=dev)
    L = u.lib()
    from ucoslam_cv3_amd._lib import check, dev_ptr

    # mapper thread: uh_ba_optimize blocks its caller between the two LM passes, so it runs where the reference runs it — on a
    # mapper thread beside the tracker (mapmanager.cpp:150 runThread) — here the worker thread of the BA object
    # (uh_ba_optimize_async / uh_ba_wait), while this thread enqueues the tracking launches
    def step():
        ba.optimize_async()
        kps, desc, counts = ext.extract_batch(frames, fp, orb_out)
        # the F frames' descriptor blocks are contiguous [F, 2000, 32]: one launch matches all F x 2000 queries against the map
        check(L.uh_knn_search_dev(index._h, dev_ptr(desc), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1))
        ba.wait()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    else:
        t_max = t_local
    counts = orb_out[2].cpu().numpy()
    full_frames = bool((counts == MAX_FEATURES).all())   # the synthetic scene yields the full 2000-keypoint budget
    total_frames = world * F * args.steps
    value = total_frames / t_max
    ms_per_step = 1e3 * t_max / args.steps

    # ---- per-stage split and roofline (rank 0; separate passes so event overhead never enters the headline number)
    roofline = None
    stage_ms = {}
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
        stage_ms["ba_ms_per_keyframe"] = timed(lambda: ba.optimize(), 5)
        # the second frame size north_star asks for: the same step (4 frames, one kNN launch, one local BA) on 640x480 frames
        fr2 = torch.from_numpy(np.stack([synth.frame(640, 480, seed=77 + f, shift=(2 * f, f)) for f in range(F)])).to(dev)
        ext2 = ORBextractor.create(ctx)
        out2 = ext2.extract_batch(fr2, fp)

        def step_640():
            ba.optimize_async()
            ext2.extract_batch(fr2, fp, out2)
            check(L.uh_knn_search_dev(index._h, dev_ptr(out2[1]), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1))
            ba.wait()

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
        del gba
        # headroom figure, NOT the metric: two independent sessions (two frame streams, two maps, two local BAs) on this one GPU —
        # the latency-bound launch chains of the two BAs interleave, which one session cannot do with itself
        ctx_ba2 = u.Context(local_rank, private=True)
        ba2 = GlobalOptimizer.create(ctx_ba2)
        ba2.setParams(synth.ba_problem(BA_K, BA_P, seed=rank + 100), ParamSet(nIters=5))
        ctx_t2 = u.Context(local_rank, private=True)
        ext_b = ORBextractor.create(ctx_t2)
        idx_b = Index(ctx_t2).build(map_desc)
        out_b = ext_b.extract_batch(frames, fp)
        knn_idx_b, knn_dist_b = torch.empty_like(knn_idx), torch.empty_like(knn_dist)

        def step_two_sessions():
            ba.optimize_async()
            ba2.optimize_async()
            ext.extract_batch(frames, fp, orb_out)
            ext_b.extract_batch(frames, fp, out_b)
            check(L.uh_knn_search_dev(index._h, dev_ptr(orb_out[1]), F * NQ, NN, dev_ptr(knn_idx), dev_ptr(knn_dist), 0, -1))
            check(L.uh_knn_search_dev(idx_b._h, dev_ptr(out_b[1]), F * NQ, NN, dev_ptr(knn_idx_b), dev_ptr(knn_dist_b), 0, -1))
            ba.wait()
            ba2.wait()

        for _ in range(3):
            step_two_sessions()
        stage_ms["two_sessions_step_ms"] = timed(step_two_sessions, 15)
        stage_ms["two_sessions_frames_per_s"] = 1e3 * 2 * F / stage_ms["two_sessions_step_ms"]
        stage_ms["orb_ms_per_frame_640x480"] = timed(lambda: ext2.extract_batch(fr2, fp, out2), 20) / F
        # not part of the metric's step (ORB + match + local BA): the per-frame pose-only solve (PnPSolver::solvePnp, 600 matches)
        from ucoslam_cv3_amd.pnp import PnPSolver

        pnp_pr = synth.pnp_problem(600, seed=3)
        pnp = PnPSolver(ctx)
        pd = {k: torch.from_numpy(np.ascontiguousarray(pnp_pr[k])).to(dev) for k in ("pose", "intr", "p3d", "kp", "invsig", "weight")}
        pwork = torch.empty(600 * 11 + 64, dtype=torch.uint8, device=dev)
        pout = (torch.empty(16, dtype=torch.float32, device=dev), torch.empty(600, dtype=torch.uint8, device=dev),
                torch.empty(5, dtype=torch.int32, device=dev), torch.empty(7, dtype=torch.float64, device=dev))
        stage_ms["pnp_ms_per_solve_600_matches"] = timed(lambda: check(L.uh_pnp_solve_dev(
            pnp._h, dev_ptr(pd["pose"]), dev_ptr(pd["intr"]), 600, dev_ptr(pd["p3d"]), dev_ptr(pd["kp"]), dev_ptr(pd["invsig"]), dev_ptr(pd["weight"]),
            dev_ptr(pwork), dev_ptr(pout[0]), dev_ptr(pout[1]), dev_ptr(pout[2]), dev_ptr(pout[3]))), 20)
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
        if not args.no_roofline:
            for c in (ctx, ctx_ba):
                c.prof_enable(True)
                c.prof_reset()
            reps = 5
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            rep = dict(ctx.prof_report())
            rep.update(ctx_ba.prof_report())
            for c in (ctx, ctx_ba):
                c.prof_enable(False)
            ab = algorithmic_bytes(F, ba_pr["E"])
            import re as _re

            short = {}
            for kname, v in rep.items():          # "(anonymous namespace)::ba_solve_kernel<true>" -> "ba_solve_kernel"
                n = _re.sub(r"<.*>", "", kname.split("::")[-1]).strip("()")
                if n == "knn_search_mq_kernel":      # the queries-per-wave form of the same search
                    n = "knn_search_kernel"
                c0, t0_ = short.get(n, (0, 0.0))
                short[n] = (c0 + v[0], t0_ + v[1])
            dom = max(short.items(), key=lambda kv: kv[1][1])
            name, (calls, tot_ms) = dom
            avg_ms = tot_ms / max(calls, 1)
            achieved = ab.get(name, 0) / (avg_ms * 1e-3) / 1e9
            traffic = None
            try:   # HBM bytes per launch from the committed PMC passes (profiles/, scripts/collect_profiles.sh)
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"].get(name)
                if pmc:
                    traffic = pmc["fetch_bytes"] + pmc["write_bytes"]
            except Exception:
                traffic = None
            roofline = {
                "bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "avg_launch_us": round(avg_ms * 1e3, 3), "algorithmic_bytes_per_launch": int(ab.get(name, 0)),
                "share_of_step_gpu_time": round(tot_ms / sum(v[1] for v in short.values()), 4),
                "kernels_us": {k: round(1e3 * v[1] / max(v[0], 1), 2) for k, v in sorted(short.items())},
                "kernels_gbps": {k: round(ab.get(k, 0) / (1e-3 * v[1] / max(v[0], 1)) / 1e9, 2) for k, v in sorted(short.items()) if ab.get(k)},
            }

    # ---- CPU baseline (rank 0, N=1 only): bounded sample of the same workload on the host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib

        O = oracle_lib.load_oracle()
        n_orb, n_match, n_ba = 12, 10, 4
        t = time.perf_counter()
        for i in range(n_orb):
            oracle_lib.orb_extract(O, frames_np[i % F], MAX_FEATURES, NLEVELS, SCALE)
        t_orb = (time.perf_counter() - t) / n_orb
        q = orb_out[1][0].cpu().numpy()
        xf = oracle_lib.load_ref("xflann")
        P = oracle_lib.P
        t = time.perf_counter()
        for i in range(n_match):
            if xf is not None:
                ii = np.empty((NQ, NN), np.int32)
                dd = np.empty((NQ, NN), np.int32)
                xf.xflann_ref_linear_search(P(map_desc_np), NT, P(q), NQ, NN, 0, 1, P(ii), P(dd))
            else:
                oracle_lib.knn_search(O, map_desc_np, q, NN, 0)
        t_match = (time.perf_counter() - t) / n_match
        g2o = oracle_lib.load_ref("g2o")
        t = time.perf_counter()
        for i in range(n_ba):
            if g2o is not None:
                oracle_lib.ba_optimize_ref(g2o, ba_pr, 5)
            else:
                oracle_lib.ba_optimize(O, ba_pr, 5)
        t_ba = (time.perf_counter() - t) / n_ba
        t_frame = t_orb + t_match + t_ba / F
        cpu = {
            "value": round(1.0 / t_frame, 4), "unit": "frames/s", "cores": 1,
            "kind": "port",
            "sample": (f"{n_orb} ORB frames (oracle port, {1e3*t_orb:.1f} ms/frame) + {n_match} matches 2000x10000 nn=10 "
                       f"({'real xflann Linear (oracle/_ref)' if xf is not None else 'oracle port'}, {1e3*t_match:.1f} ms) + {n_ba} local BAs "
                       f"({'real g2o (oracle/_ref)' if g2o is not None else 'oracle port'}, {1e3*t_ba:.1f} ms), one BA per {F} frames, single thread"),
        }

    if rank == 0:
        line = {
            "metric": "tracking frames/sec (ORB extract + match + local BA), 1241x376 mono",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (ORB, Hamming) + f64 (BA)", "data": "synthetic",
            "config": {"workload": "orb1241x376_2000f_8lv + hamming_knn_2000x10000_nn10 + local_ba_10kf_3000pt",
                       "frames_per_step": F, "frames_per_keyframe": F, "parallelism": f"frame-streams x{world} (replicas, no data-path collective); tracking and local-BA on two HIP streams per GPU"},
            "stages": {k: round(v, 4) for k, v in stage_ms.items()}, "keypoints_per_frame": int(counts.min()), "full_budget": full_frames,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 2)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which a pipe buffers until exit: drain it first so that the JSON
        # line is the LAST line on stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
