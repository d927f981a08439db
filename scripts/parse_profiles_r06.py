"""Condenses gpurun_out/prof_r06 (scripts/collect_profiles_r06.sh) into the tracked summaries under profiles/:
  r06_quick_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --quick` (the headline loop only)
  r06_tracker_kernel_stats.csv   the same of the tracker's per-frame chain (examples/tracker_frame.cpp)
  r06_tracker_frame.json         that program's own per-stage latencies, un-profiled
  r06_pmc_traffic.json           per-kernel HBM traffic per launch from separate FETCH_SIZE / WRITE_SIZE passes over the headline loop
  r06_chain_kernel_stats.csv     kernel stats of the local-BA launch chain at 17 / 24 / 32 / 48 / 64 free keyframes (ba_window_sweep.py)
  r06_chain_mfma.json            per chain kernel: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (MFMA pipe busy fraction of the time any wave is
                                 resident), f64 MFMA ops per launch, traffic per launch
Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE count KiB of L2 <-> fabric requests; on gfx950 FETCH_SIZE reports half the
bytes of a wide (16 B / lane) coalesced stream, so `fetch_bytes_x2` is given beside the raw value."""
import collections, csv, glob, json, os, re, shutil

src = os.path.join("gpurun_out", "prof_r06")
os.makedirs("profiles", exist_ok=True)


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[0] if g else None


for sub, name in (("quick", "r06_quick_kernel_stats.csv"), ("tracker", "r06_tracker_kernel_stats.csv"), ("tracker_dev", "r06_tracker_dev_frame_kernel_stats.csv"), ("tracker_fused", "r06_tracker_fused_kernel_stats.csv"), ("hkmeans", "r06_hkmeans_kernel_stats.csv"), ("chain", "r06_chain_kernel_stats.csv")):
    f = first(os.path.join(src, sub, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join("profiles", name))
        print("copied", f, "->", name)
for nm, dst in (("tracker_fused_plain.json", "r06_tracker_frame_fused.json"),):
    tpf = os.path.join(src, nm)
    if os.path.exists(tpf):
        line = [l for l in open(tpf).read().splitlines() if l.startswith("{")]
        if line:
            json.dump(json.loads(line[-1]), open(os.path.join("profiles", dst), "w"), indent=1)
tl = os.path.join(src, "tracker_fused_timeline.txt")
if os.path.exists(tl):
    shutil.copy(tl, os.path.join("profiles", "r06_tracker_fused_timeline.txt"))
tp = os.path.join(src, "tracker_plain.json")
if os.path.exists(tp):
    line = [l for l in open(tp).read().splitlines() if l.startswith("{")]
    if line:
        json.dump(json.loads(line[-1]), open(os.path.join("profiles", "r06_tracker_frame.json"), "w"), indent=1)


def agg(path, counters):
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not path:
        return d
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in counters:
            continue
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        e = d[k][r["Counter_Name"]]
        e[0] += 1; e[1] += float(r["Counter_Value"])
    return d


f = agg(first(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv")), {"FETCH_SIZE"})
w = agg(first(os.path.join(src, "pmc_write", "**", "*counter_collection.csv")), {"WRITE_SIZE"})
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --quick --steps 6 --warmup 2 --reps 1   (the headline step: host in / host out, a fresh BA problem per keyframe)",
       "units": "bytes per launch (counter KiB * 1024); fetch_bytes_x2 applies the gfx950 wide-read correction", "kernels": {}}
for k in sorted(set(f) | set(w)):
    fn, fv = f[k]["FETCH_SIZE"] if k in f else (0, 0.0)
    wn, wv = w[k]["WRITE_SIZE"] if k in w else (0, 0.0)
    fb, wb = 1024 * fv / max(fn, 1), 1024 * wv / max(wn, 1)
    out["kernels"][k] = {"launches_sampled": fn, "fetch_bytes": round(fb), "fetch_bytes_x2": round(2 * fb), "write_bytes": round(wb)}
json.dump(out, open(os.path.join("profiles", "r06_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if k.startswith(("ba_", "knn_stream", "cell_nms"))}, indent=0)[:1500])


# ---- the launch chain's kernels (17 / 24 / 32 / 48 / 64 free keyframes): MFMA busy fraction, f64 MFMA ops, traffic
mf = agg(first(os.path.join(src, "chain_mfma", "**", "*counter_collection.csv")), {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"})
mo = agg(first(os.path.join(src, "chain_mops", "**", "*counter_collection.csv")), {"SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_WAVE_CYCLES"})
cf = agg(first(os.path.join(src, "chain_fetch", "**", "*counter_collection.csv")), {"FETCH_SIZE"})
cw = agg(first(os.path.join(src, "chain_write", "**", "*counter_collection.csv")), {"WRITE_SIZE"})
dur = {}
ks = first(os.path.join(src, "chain", "**", "*kernel_stats.csv"))
if ks:
    for r in csv.DictReader(open(ks)):
        m = re.search(r"(\w+_kernel)", r["Name"])
        k = m.group(1) if m else r["Name"][:40]
        e = dur.setdefault(k, [0, 0.0])
        e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"])
chain = {"command": "rocprofv3 --pmc <counters> --kernel-trace -- python scripts/ba_window_sweep.py 3000 17 24 32 48 64   (UH_SWEEP_NO_ORACLE=1; one pass per counter set)",
         "note": "mfma_util_of_chip = SQ_VALU_MFMA_BUSY_CYCLES per launch / (average launch duration x 2.4 GHz x 256 CUs x 4 SIMDs): the MfmaUtil formula of gfx94x with the kernel's duration as GRBM_GUI_ACTIVE — the share of the chip's matrix-pipe cycles the kernel keeps busy (all windows 17-64 free keyframes pooled).  v_mfma_f64_16x16x4_f64 sustains 6.4 FMA / clock / SIMD on gfx950 (profiles/r03_mfma_f64.json) against 11.4 for v_fma_f64 (scripts/micro/chain_latency.hip): these kernels are bound by the instructions around the MFMAs and by launch / memory latency, not by the matrix pipe.",
         "kernels": {}}
for k in sorted(set(mf) | set(mo) | set(cf) | set(cw)):
    if not k.startswith(("ba_", "uh_")):
        continue
    e = {}
    if k in dur:
        e["launches"] = dur[k][0]; e["avg_us"] = round(dur[k][1] / max(dur[k][0], 1) / 1e3, 2)
    if k in mf:
        b, s_ = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1], mf[k]["SQ_BUSY_CYCLES"][1]
        e["mfma_busy_cycles_per_launch"] = round(b / max(mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0], 1)); e["sq_busy_cycles_per_launch"] = round(s_ / max(mf[k]["SQ_BUSY_CYCLES"][0], 1))
        # (no busy-over-busy ratio: SQ_BUSY_CYCLES is per shader engine, SQ_VALU_MFMA_BUSY_CYCLES per SIMD — their quotient exceeded 1 and is not a utilisation)
        if k in dur and dur[k][0]:
            # gfx94x MfmaUtil formula with the kernel's own duration for GRBM_GUI_ACTIVE: busy SIMD-cycles / (duration x 2.4 GHz x 256 CUs x 4 SIMDs)
            e["mfma_util_of_chip"] = round((b / max(mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0], 1)) / (dur[k][1] / dur[k][0] * 2.4 * 256 * 4), 5)
    if k in mo:
        e["mfma_mops_f64_per_launch"] = round(mo[k]["SQ_INSTS_VALU_MFMA_MOPS_F64"][1] / max(mo[k]["SQ_INSTS_VALU_MFMA_MOPS_F64"][0], 1))
    if k in cf:
        e["fetch_bytes_per_launch"] = round(1024 * cf[k]["FETCH_SIZE"][1] / max(cf[k]["FETCH_SIZE"][0], 1))
    if k in cw:
        e["write_bytes_per_launch"] = round(1024 * cw[k]["WRITE_SIZE"][1] / max(cw[k]["WRITE_SIZE"][0], 1))
    chain["kernels"][k] = e
if chain["kernels"]:   # (the chain part is collected only with UH_COLLECT_PARTS=... chain)
    json.dump(chain, open(os.path.join("profiles", "r06_chain_mfma.json"), "w"), indent=1)
print(json.dumps(chain["kernels"], indent=0)[:3000])
