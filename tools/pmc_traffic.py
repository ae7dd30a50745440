"""profiles/r01_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) of
   python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of a
wide coalesced read -> fetch_bytes_corrected = 2 * FETCH_SIZE * 1024."""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(root, counter):
    acc = defaultdict(list)
    wgs = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                n_wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
                key = "%s wgs=%d" % (r["Kernel_Name"], n_wg)
                acc[key].append(float(r["Counter_Value"]))
                wgs[key] = n_wg
    return acc, wgs


fetch, wgs = collect(sys.argv[1], "FETCH_SIZE")
write, _ = collect(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and WRITE_SIZE in a separate pass) -- python bench.py --steps 8 "
               "--warmup 2 --no-cpu-baseline --no-batch32 --no-graph ; FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 correction: "
               "FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> fetch_bytes_corrected "
               "= 2*FETCH_SIZE*1024 (tools/pmc_traffic.py)", "kernels": {}}
for k in sorted(fetch):
    if not any(s in k for s in ("qmm_", "paged_attn", "argmax", "embedding")):
        continue
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write[k]) / len(write[k]) if k in write else 0.0
    out["kernels"][k] = {"workgroups": wgs[k], "launches": len(fetch[k]), "FETCH_SIZE_KiB_avg": round(f, 1),
                         "fetch_bytes_corrected": int(2 * f * 1024), "WRITE_SIZE_KiB_avg": round(w, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"].get("void qmm_kernel<1, 2, 12>(QmmArgs) wgs=896"), indent=1))
