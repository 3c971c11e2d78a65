"""Summarises the PMC passes written by tools/pmc.sh: per kernel, summed over dispatches."""
import collections, csv, glob, sys
name = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
p5 = collections.defaultdict(lambda: collections.defaultdict(float))     # the lane-utilisation pass on its own: its ratios must come from ONE pass
for f in sorted(glob.glob(f"gpurun_out/{name}/p*/p_counter_collection.csv")):
    last = "/p5/" in f
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
        if last:
            p5[k][r["Counter_Name"]] += float(r["Counter_Value"])
        else:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if not k.startswith("k_"): continue
    print(k)
    for a in sorted(v): print(f"    {a:44s} {v[a]:.4g}")
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if g:
        print(f"    -> cycles/XCD {g:.4g}; TA busy {v['TA_TA_BUSY_sum']/256/g:.2%}; TD busy {v['TD_TD_BUSY_sum']/256/g:.2%}; "
              f"VALU busy {v['SQ_ACTIVE_INST_VALU']*4/1024/g:.2%}; L2 hit {v['TCC_HIT_sum']/max(v['TCC_REQ_sum'],1):.2%}; "
              f"VMEM_RD/wave {v['SQ_INSTS_VMEM_RD']/max(v['SQ_WAVES'],1):.1f}; VALU/wave {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):.1f}")
    q = p5.get(k)
    if q and q.get("SQ_ACTIVE_INST_VALU"):
        g5 = q.get("GRBM_GUI_ACTIVE", 0) / 8
        print(f"    -> live lanes per busy VALU cycle {q['SQ_THREAD_CYCLES_VALU']/q['SQ_ACTIVE_INST_VALU']:.1f} of 64 "
              f"(lane utilisation {q['SQ_THREAD_CYCLES_VALU']/q['SQ_ACTIVE_INST_VALU']/64:.2%}); "
              f"wave-cycles on scalar instructions {q['SQ_ACTIVE_INST_SCA']*4/1024/g5 if g5 else 0:.2%} of SIMD-cycles (SQ_ACTIVE_INST_SCA x 4 / 1024 SIMDs), "
              f"scalar ALU instructions per CU-cycle {q['SQ_INSTS_SALU']/256/g5 if g5 else 0:.2f}")
