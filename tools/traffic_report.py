"""Turns the two passes of tools/traffic.sh into profiles/traffic.json: HBM-side bytes per launch of each kernel.

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 bytes, so the
read side is doubled (an upper bound for narrow requests); both the raw and the corrected figures are kept.

    python tools/traffic_report.py <outdir-name> <frames> [cells] > profiles/traffic.json
"""
import collections, csv, glob, json, sys

name, frames = sys.argv[1], int(sys.argv[2])
cells = int(sys.argv[3]) if len(sys.argv) > 3 else 708
agg = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(f"gpurun_out/{name}/{which}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            agg[k][counter] += float(r["Counter_Value"])
            if which == "fetch":
                agg[k]["launches"] += 1
kernels = {}
for k, v in agg.items():
    n = max(v["launches"], 1)
    kernels[k] = {
        "launches": v["launches"], "launches_per_frame": v["launches"] / frames,
        "fetch_bytes_raw_per_launch": v["FETCH_SIZE"] * 1024 / n, "write_bytes_per_launch": v["WRITE_SIZE"] * 1024 / n,
        "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n,
    }
# k_shadow comes in variants (with and without a walk budget); a frame's launches are counted together
sh = [v for k, v in agg.items() if k.startswith("k_shadow")]
sh_launches = sum(v["launches"] for v in sh)
sh_bytes = sum((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 for v in sh)
out = {"kernel": "k_shadow", "n_gpus": 1, "workload_cells": cells, "frames": frames,
       "hbm_bytes_per_launch": sh_bytes / sh_launches if sh_launches else None,
       "correction": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024; FETCH_SIZE doubled for gfx950's 64-byte tally of 128-byte requests",
       "kernels": kernels}
print(json.dumps(out, indent=1))
