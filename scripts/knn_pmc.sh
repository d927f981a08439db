#!/bin/bash
# Runs on the GPU box (via gpurun): hardware counters of the exact kNN kernel (8000 x 10000, nn = 2 and 10), one rocprofv3 pass per
# counter group (no trace domains besides --kernel-trace).  Output: gpurun_out/knn_pmc/<group>/..._counter_collection.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/knn_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/time_knn.py"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES" "TCC_HIT TCC_MISS TCC_REQ TCC_BUSY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d $OUT/g$i -o knn -- $CMD > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json, re
out = {"what": "rocprofv3 --pmc passes over scripts/time_knn.py (exact kNN, 10000 train rows; 2000 queries: fused kernel, 8000 queries: accept scan + lane replay), per-launch averages per kernel; scripts/knn_pmc.sh", "kernels": {}}
for g in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(g + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            m = re.search(r"(knn_\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
            if not m: continue
            key = m.group(1)
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
        for key in acc:
            out["kernels"].setdefault(key, {}).update({c: round(v / n[(key, c)]) for c, v in acc[key].items()})
            out["kernels"][key]["launches_sampled"] = max(n[(key, c)] for c in acc[key])
json.dump(out, open("$OUT/knn_pmc.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=0)[:3000])
PY
