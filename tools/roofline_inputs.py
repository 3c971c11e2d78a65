"""Per-launch counter figures of the dominant kernel for bench.py's roofline, from the passes of tools/pmc.sh and
tools/traffic.sh over tools/prof_frame.py (default workload: BASELINE config 4).

    python tools/roofline_inputs.py <pmc-outdir-name> <traffic-outdir-name> > profiles/r3_roofline_inputs.json

The file carries a hash of the kernel sources it was measured on (raytracer_amd._native.source_hash, which bench.py repeats): counters of
another build of the kernels are not this build's roofline.

valu_busy_cycles_per_launch = SQ_ACTIVE_INST_VALU x 4 (the counter ticks in quad-cycles; rocprofv3 sums it over the
chip); hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE doubled because gfx950 tallies its
128-byte fabric requests at 64 bytes (an upper bound for narrower requests).
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd._native import source_hash  # noqa: E402

pmc, traffic = sys.argv[1], sys.argv[2]
KERNEL = "k_shadow<false, false, false>"   # (not counted, no budget, rays without 1/direction: the variant config 4 runs)
agg = collections.defaultdict(float)
launches = collections.defaultdict(int)
lane = collections.defaultdict(float)      # the lane-utilisation pass (p5) by itself: its ratios come from one pass
for f in glob.glob(f"gpurun_out/{pmc}/p*/*counter_collection.csv") + glob.glob(f"gpurun_out/{traffic}/*/**/*counter_collection.csv", recursive=True):
    p5 = "/p5/" in f
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        if p5:
            lane[r["Counter_Name"]] += float(r["Counter_Value"])
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Counter_Name"]] += 1
per = {k: agg[k] / launches[k] for k in agg}
cyc_xcd = per.get("GRBM_GUI_ACTIVE", 0) / 8
out = {
    "workload_cells": 708, "resolution": [3840, 2160], "lights": 8, "n_gpus": 1,
    "view": "frame ROOFLINE_FRAME of bench.py's camera path (tools/prof_frame.py)",
    "code_sha256": source_hash(),
    "source": f"tools/pmc.sh {pmc} + tools/traffic.sh {traffic} (rocprofv3 --pmc, kernels serialised by the profiler)",
    "k_shadow_main": {
        "kernel": KERNEL, "launches_profiled": launches.get("SQ_WAVES", 0),
        "valu_busy_cycles_per_launch": per.get("SQ_ACTIVE_INST_VALU", 0) * 4,
        "valu_insts_per_launch": per.get("SQ_INSTS_VALU"),
        "salu_insts_per_launch": per.get("SQ_INSTS_SALU"), "smem_insts_per_launch": per.get("SQ_INSTS_SMEM"),
        "waves_per_launch": per.get("SQ_WAVES"),
        "wave_cycles_per_launch": per.get("SQ_WAVE_CYCLES", 0) * 4, "wait_any_cycles_per_launch": per.get("SQ_WAIT_ANY", 0) * 4,
        "gui_active_cycles_per_xcd_per_launch": cyc_xcd,
        "valu_busy_frac_measured_clock": per.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc_xcd if cyc_xcd else None,
        "l2_hit": per.get("TCC_HIT_sum", 0) / per["TCC_REQ_sum"] if per.get("TCC_REQ_sum") else None,
        # of the cycles the vector ALUs are busy, the share of their 64 lanes that is live (SQ_THREAD_CYCLES_VALU /
        # (SQ_ACTIVE_INST_VALU x 64), one counter pass); the scalar unit's busy cycles per CU-cycle
        "valu_lane_utilisation": lane["SQ_THREAD_CYCLES_VALU"] / lane["SQ_ACTIVE_INST_VALU"] / 64 if lane.get("SQ_ACTIVE_INST_VALU") else None,
        # the scalar side: wave-cycles spent on scalar ALU / scalar memory instructions per SIMD-cycle (SQ_ACTIVE_INST_SCA x 4
        # over 1024 SIMDs x the XCD's busy cycles: a SIMD issues for one wave at a time), and scalar ALU instructions per
        # CU-cycle (a CU's one scalar unit takes at most one per cycle)
        "salu_busy": lane["SQ_ACTIVE_INST_SCA"] * 4 / 1024 / (lane["GRBM_GUI_ACTIVE"] / 8) if lane.get("GRBM_GUI_ACTIVE") else None,
        "salu_insts_per_cu_cycle": lane["SQ_INSTS_SALU"] / 256 / (lane["GRBM_GUI_ACTIVE"] / 8) if lane.get("GRBM_GUI_ACTIVE") else None,
        "fetch_size_raw_bytes_per_launch": per.get("FETCH_SIZE", 0) * 1024, "write_bytes_per_launch": per.get("WRITE_SIZE", 0) * 1024,
        "hbm_bytes_per_launch": (2 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024,
    },
}
print(json.dumps(out, indent=1))
