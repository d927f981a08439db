#!/bin/bash
# Runs on the GPU box (via gpurun): hardware counters of the local-BA kernels (10 keyframes x 3000 landmarks), one rocprofv3 pass per
# counter group (no trace domains besides --kernel-trace).  Prints per-kernel per-launch averages; raw CSVs under gpurun_out/ba_pmc/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ba_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/time_ba.py"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d $OUT/g$i -o ba -- $CMD > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json, re
res = {}
for g in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(g + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            m = re.search(r"(ba_[a-z_]+kernel(<\w+>)?)", r["Kernel_Name"])
            if not m: continue
            key = m.group(1)
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
        for key in acc:
            res.setdefault(key, {}).update({c: round(v / n[(key, c)], 1) for c, v in acc[key].items()})
json.dump({"what": "rocprofv3 --pmc passes over scripts/time_ba.py (local BA 10 x 3000), per-launch averages incl. the no-op launches of finished passes; scripts/ba_pmc.sh", "kernels": res}, open("$OUT/summary.json", "w"), indent=1)
for k in ("ba_backsub_kernel<true>", "ba_schur_kernel", "ba_lin_kernel"):
    print(k, res.get(k))
PY
