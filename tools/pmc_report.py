"""Summarises the PMC passes written by tools/pmc.sh: per kernel, summed over dispatches."""
import collections, csv, glob, sys
name = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/{name}/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if not k.startswith("k_"): continue
    print(k)
    for a in sorted(v): print(f"    {a:44s} {v[a]:.4g}")
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if g:
        print(f"    -> cycles/XCD {g:.4g}; TA busy {v['TA_TA_BUSY_sum']/256/g:.2%}; TD busy {v['TD_TD_BUSY_sum']/256/g:.2%}; "
              f"VALU busy {v['SQ_ACTIVE_INST_VALU']*4/1024/g:.2%}; L2 hit {v['TCC_HIT_sum']/max(v['TCC_REQ_sum'],1):.2%}; "
              f"VMEM_RD/wave {v['SQ_INSTS_VMEM_RD']/max(v['SQ_WAVES'],1):.1f}; VALU/wave {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):.1f}")
