"""FETCH_SIZE of tools/fetch_calib.hip's kernels beside the bytes each of them read.

    python tools/fetch_calib_report.py <counter dir> <program's stdout file> > profiles/r2_fetch_calibration.json

FETCH_SIZE is in KiB.  `ratio` = bytes read / (FETCH_SIZE x 1024): 2 means the counter tallies the pattern at half, as it
is known to do for 16-byte-per-lane streaming reads on gfx950; 1 means it is exact for the pattern.
"""
import csv, glob, json, sys

cdir, out = sys.argv[1], sys.argv[2]
read = json.loads([l for l in open(out) if l.startswith("{")][-1])
fetch = {}
for f in glob.glob(cdir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            fetch[k] = fetch.get(k, 0.0) + float(r["Counter_Value"])
rows = {}
for k, b in read.items():
    kib = fetch.get(k)
    rows[k] = {"bytes_read": b, "FETCH_SIZE_KiB": kib, "ratio": (b / (kib * 1024.0)) if kib else None}
print(json.dumps({"buffer": "2 GiB, every byte read once per kernel (lane_records<5u>: 80 of every 80 bytes of the first 1.25 GiB)",
                  "patterns": rows, "kernels_seen": sorted(fetch)}, indent=1))
