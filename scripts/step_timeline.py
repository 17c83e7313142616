"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV: start, end, duration (us, relative to the end of
a backward compositor launch), queue, kernel, grid.  usage: step_timeline.py <kernel_trace.csv> [which step from the end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = [i for i, r in enumerate(rows) if "raster_bwd_kernel" in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    n = r["Kernel_Name"].replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::").replace("void ", "")
    m = re.search(r"wrapped_(\w+)", n)
    n = (m.group(0) if m else re.sub(r"[(].*", "", n))[:90]
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{r['Queue_Id']} {n}  g{r['Grid_Size_X']}")
