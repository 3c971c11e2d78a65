"""Per-launch lane utilisation from one rocprofv3 counter pass.

usage: pmc_per_launch.py <dir with *_counter_collection.csv>
The pass is
    rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv \
              -d <dir> -o p -- python3 tools/prof_scene.py huge_grid 3840 2160 2
Prints, for every launch of the last frame in launch order: the kernel, its VALU-busy cycles, vector instructions, waves,
and the share of lanes that were live while the vector ALUs were busy (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64);
k_primary, whose lanes are all live, reads 1.0 and checks the scale)."""
import collections
import csv
import glob
import os
import sys

files = glob.glob(os.path.join(sys.argv[1], "*counter_collection.csv"))
rows = list(csv.DictReader(open(files[0])))
launches = collections.OrderedDict()
for r in rows:
    e = launches.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0][:44]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.) + float(r["Counter_Value"])
order = list(launches.values())
last = [i for i, v in enumerate(order) if v["name"].startswith("k_primary")][-1]
print(f"{'kernel':46s} {'VALU-busy Mcycles':>18s} {'VALU insts M':>13s} {'waves':>8s} {'live lanes':>11s}")
for v in order[last:]:
    busy = v.get("SQ_ACTIVE_INST_VALU", 0.)
    if busy <= 0:
        continue
    print(f"{v['name']:46s} {busy * 4 / 1e6:18.1f} {v.get('SQ_INSTS_VALU', 0.) / 1e6:13.1f} {int(v.get('SQ_WAVES', 0)):8d} "
          f"{v.get('SQ_THREAD_CYCLES_VALU', 0.) / (busy * 64):11.3f}")
