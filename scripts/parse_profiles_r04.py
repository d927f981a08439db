"""Condenses gpurun_out/prof_r04 (scripts/collect_profiles_r04.sh) into the tracked summaries under profiles/:
  r04_quick_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --quick` (the headline loop only)
  r04_tracker_kernel_stats.csv   the same of the tracker's per-frame chain (examples/tracker_frame.cpp)
  r04_tracker_frame.json         that program's own per-stage latencies, un-profiled
  r04_pmc_traffic.json           per-kernel HBM traffic per launch from separate FETCH_SIZE / WRITE_SIZE passes over the headline loop
Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE count KiB of L2 <-> fabric requests; on gfx950 FETCH_SIZE reports half the
bytes of a wide (16 B / lane) coalesced stream, so `fetch_bytes_x2` is given beside the raw value."""
import collections, csv, glob, json, os, re, shutil

src = os.path.join("gpurun_out", "prof_r04")
os.makedirs("profiles", exist_ok=True)


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[0] if g else None


for sub, name in (("quick", "r04_quick_kernel_stats.csv"), ("tracker", "r04_tracker_kernel_stats.csv")):
    f = first(os.path.join(src, sub, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join("profiles", name))
        print("copied", f, "->", name)
tp = os.path.join(src, "tracker_plain.json")
if os.path.exists(tp):
    line = [l for l in open(tp).read().splitlines() if l.startswith("{")]
    if line:
        json.dump(json.loads(line[-1]), open(os.path.join("profiles", "r04_tracker_frame.json"), "w"), indent=1)


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
json.dump(out, open(os.path.join("profiles", "r04_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if k.startswith(("ba_", "knn_stream", "cell_nms"))}, indent=0)[:1500])
