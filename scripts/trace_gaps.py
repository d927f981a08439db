"""Kernel timeline of one steady-state frame from a rocprofv3 kernel trace (csv): name, start offset, duration, gap to the previous kernel."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "ingest"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0, i1 = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print(f"{name:44s} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}")
    prev_end = e
print(f"frame span (first start to next frame's first start): {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
