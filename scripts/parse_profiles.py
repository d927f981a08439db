"""Condenses gpurun_out/prof_rNN (scripts/collect_profiles.sh) into the tracked summaries under profiles/:
  profiles/rNN_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3`
  profiles/rNN_pmc_traffic.json        per-kernel HBM traffic per launch from separate FETCH_SIZE / WRITE_SIZE passes
Units (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB of L2<->fabric requests; on gfx950 FETCH_SIZE reports half
the bytes of a wide (16 B/lane) coalesced stream, other access widths are uncalibrated, so `fetch_bytes_x2` is given next to
the raw value and bench.py reports raw fetch + write as `traffic` with that caveat."""
import collections, csv, json, os, re, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", f"prof_{rnd}")
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join("profiles", f"{rnd}_bench_kernel_stats.csv"))


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"::(\w+_kernel)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


f = agg(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"), "FETCH_SIZE")
w = agg(os.path.join(src, "pmc_write", "bench_counter_collection.csv"), "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline",
       "units": "bytes per launch (counter KiB * 1024); fetch_bytes_x2 applies the gfx950 wide-read correction", "kernels": {}}
for k in sorted(set(f) | set(w)):
    fb = 1024 * f[k][1] / max(f[k][0], 1)
    wb = 1024 * w[k][1] / max(w[k][0], 1)
    out["kernels"][k] = {"launches_sampled": f[k][0], "fetch_bytes": round(fb), "fetch_bytes_x2": round(2 * fb), "write_bytes": round(wb)}
json.dump(out, open(os.path.join("profiles", f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=0)[:1500])
