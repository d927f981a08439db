"""Condenses gpurun_out/prof_r03 (scripts/collect_profiles_r03.sh) into the tracked summaries under profiles/:
  r03_quick_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --quick` (the headline loop only)
  r03_bench_kernel_stats.csv   the same of the full bench command
  r03_sharded_kernel_stats.csv the sharded frame stream with one rank
  r03_pmc_traffic.json         per-kernel HBM traffic per launch from separate FETCH_SIZE / WRITE_SIZE passes over the headline loop
  r03_mfma_f64.json            fp64 MFMA vs vector FMA: micro-benchmark issue rates + MFMA-busy / VALU counters of ba_persist_kernel per form
Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE count KiB of L2 <-> fabric requests; on gfx950 FETCH_SIZE reports half the
bytes of a wide (16 B / lane) coalesced stream, so `fetch_bytes_x2` is given beside the raw value (the BA kernel's exchange is 16-byte
tagged words: the x2 figure is the one that applies to it)."""
import collections, csv, glob, json, os, re, shutil

src = os.path.join("gpurun_out", "prof_r03")
os.makedirs("profiles", exist_ok=True)


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[0] if g else None


for sub, name in (("quick", "r03_quick_kernel_stats.csv"), ("stats", "r03_bench_kernel_stats.csv"), ("shard", "r03_sharded_kernel_stats.csv")):
    f = first(os.path.join(src, sub, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join("profiles", name))
        print("copied", f, "->", name)


def agg(path, counters):
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not path:
        return d
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in counters:
            continue
        m = re.search(r"::(\w+_kernel)", r["Kernel_Name"])
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
json.dump(out, open(os.path.join("profiles", "r03_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if k.startswith(("ba_", "knn_stream", "cell_nms"))}, indent=0)[:1200])

mf = {"what": "fp64 MFMA against vector FMA on gfx950 (north_star: 'MFMA ... evidenced by rocprof'): scripts/micro/mfma_f64_rate.hip issue rates, and hardware counters of "
              "ba_persist_kernel<8> (local BA 10 x 3000, scripts/time_ba.py) with its Schur product as v_mfma_f64_16x16x4_f64 (UH_BA_SCHUR=mfma) and as register-blocked v_fma_f64 (default)",
      "micro_benchmark": open(os.path.join(src, "mfma", "rate.txt")).read().strip().split("\n") if os.path.exists(os.path.join(src, "mfma", "rate.txt")) else None,
      "counters_available": open(os.path.join(src, "mfma", "counters_available.txt")).read().strip().split("\n")[:40] if os.path.exists(os.path.join(src, "mfma", "counters_available.txt")) else None,
      "forms": {}}
for form in ("mfma", "valu"):
    ent = {}
    for sub in ("pmc_", "pmc2_"):
        a = agg(first(os.path.join(src, "mfma", sub + form, "**", "*counter_collection.csv")),
                {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VALU"})
        for k, cs in a.items():
            if k.startswith("ba_persist"):
                ent.update({c: round(v[1] / max(v[0], 1), 1) for c, v in cs.items()})
    t = os.path.join(src, "mfma", f"time_{form}.txt")
    if os.path.exists(t):
        ent["optimize_ms_line"] = open(t).read().strip()
    if ent.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and ent.get("SQ_BUSY_CYCLES"):
        ent["mfma_busy_over_sq_busy"] = round(ent["SQ_VALU_MFMA_BUSY_CYCLES"] / ent["SQ_BUSY_CYCLES"], 4)
    mf["forms"][form] = ent
json.dump(mf, open(os.path.join("profiles", "r03_mfma_f64.json"), "w"), indent=1)
print(json.dumps(mf["forms"], indent=0))
