#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/knn_pmc5
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for nq in 8000 2000; do
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d $OUT/q$nq/g$i -o knn -- python $R/scripts/knn_one_frame.py $nq > $OUT/q${nq}_g$i.log 2>&1
done
done
python - <<PY
import csv, glob, collections, json, re
for nq in (8000, 2000):
    out = {}
    for f in glob.glob("$OUT/q%d/g*/**/*counter_collection.csv" % nq, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "knn_stream" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for c in acc: out[c] = round(acc[c] / n[c])
    print(nq, json.dumps(out))
PY
