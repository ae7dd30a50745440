"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) of
   python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [commit]
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
out["commit"] = sys.argv[4] if len(sys.argv) > 4 else "unknown"
# the counters describe the kernels AS BUILT FROM THESE SOURCES: bench.py quotes `dominant` only while the hash still matches
import hashlib, os
_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "candle_vllm_amd", "csrc")
_h = hashlib.sha256()
for _f in sorted(os.listdir(_csrc)):
    if _f.startswith("qmatmul") or _f.startswith("qmm_") or _f == "common.h":
        _h.update(open(os.path.join(_csrc, _f), "rb").read())
out["kernel_source_sha256"] = _h.hexdigest()
# the dominant launch group of the step = the mat-vec kernel that moves the most bytes per step (gate/up at batch 1):
# recorded BY MEASUREMENT so that the bench needs no kernel name
mv = {k: v for k, v in out["kernels"].items() if "qmm_kernel" in k and v["launches"] > 0}
if mv:
    dom = max(mv, key=lambda k: mv[k]["fetch_bytes_corrected"] * mv[k]["launches"])
    e = mv[dom]
    out["dominant"] = {"kernel": dom, "workgroups": e["workgroups"], "launches": e["launches"],
                       "traffic_bytes_per_launch": int(e["fetch_bytes_corrected"] + e["WRITE_SIZE_KiB_avg"] * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out.get("dominant"), indent=1))
