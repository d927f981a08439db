"""From a rocprofv3 kernel trace of scripts/time_kdbuild.py: duration of the single-launch builds and the span of the three-launch builds
(start of kd_build_kernel<true> to end of kd_join_kernel)."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
i = 0
while i < len(rows):
    k = rows[i]["Kernel_Name"]
    if "kd_build_kernel<true>" in k or "kd_build_kernelILb1" in k:
        j = i
        while j < len(rows) and "kd_join_kernel" not in rows[j]["Kernel_Name"]:
            j += 1
        parts = [(r["Kernel_Name"].split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[i:j + 1]]
        span = (int(rows[j]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        out.append(("split", span, parts))
        i = j + 1
        continue
    if "kd_build_kernel" in k:
        out.append(("single", (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, []))
    i += 1
for kind, us, parts in out:
    print(f"{kind:7s} {us:8.1f} us  " + "  ".join(f"{n}={d:.1f}" for n, d in parts))
