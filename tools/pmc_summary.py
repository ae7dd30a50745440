"""aggregate a rocprofv3 --pmc counter_collection csv: per (kernel, grid) mean of every counter -> json"""
import csv
import glob
import json
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            key = "%s grid=%s wg=%s" % (r.get("Kernel_Name", "?")[:90], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    res[k] = {"launches": max(len(v) for v in cs.values())}
    for c, v in cs.items():
        res[k][c] = round(sum(v) / len(v), 1)
if len(sys.argv) > 3:
    res = {"commit": sys.argv[3], "kernels": res}
    json.dump(res, open(out, "w"), indent=1)
    res = res["kernels"]
else:
    json.dump(res, open(out, "w"), indent=1)
for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0) * res[k]["launches"]):
    print(k, res[k])
