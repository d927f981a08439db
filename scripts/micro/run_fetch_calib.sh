#!/bin/bash
# FETCH_SIZE calibration on the GPU box; result condensed into profiles/r05_fetch_size_calibration.json by the python below
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/fetch_calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT -o calib -- $R/scripts/micro/fetch_calib.bin > $OUT/run.log 2>&1
python - <<PY
import csv, glob, json
f = sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True))
rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE"] if f else []
out = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    out.append({"kernel": name, "dispatch": int(r["Dispatch_Id"]), "fetch_size_KiB": float(r["Counter_Value"]), "fetch_bytes": float(r["Counter_Value"]) * 1024})
json.dump(out, open("$OUT/calib.json", "w"), indent=1)
print(json.dumps(out, indent=0))
PY
tail -2 $OUT/run.log
