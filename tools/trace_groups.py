"""scratch: group a rocprofv3 kernel-trace CSV by (kernel, grid) -> calls, mean / min duration"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0][-48:]
        grid = (r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        rows[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
for (name, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 0.002 * tot:
        continue
    print(f"{name:50s} grid={grid!s:28s} n={len(v):5d} mean={sum(v) / len(v) / 1e3:8.2f} us min={min(v) / 1e3:8.2f} share={sum(v) / tot:.3f}")
