"""Per-kernel time per frame from a tools/ktrace.sh run:  python tools/kt_summary.py <outdir-name> <frames>"""
import csv, sys
name, frames = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(f"gpurun_out/{name}/kt_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if float(r["TotalDurationNs"]) / tot < 0.004: continue
    print(f"{r['Name'][:58]:58s} calls {r['Calls']:>4s} ms/frame {float(r['TotalDurationNs'])/1e6/frames:7.3f} avg {float(r['AverageNs'])/1e3:8.1f} us")
print(f"total ms/frame {tot/1e6/frames:.3f}")
