#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $OUT/a -o ba -- python $R/scripts/time_ba.py > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace -f csv -d $OUT/b -o ba -- python $R/scripts/time_ba.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ('a','b'):
    for f in glob.glob(f"/root/repo/gpurun_out/prof_lds/{d}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'ba_persist' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d, k, sum(v)/len(v), len(v))
PY
