import json,sys
for p in sys.argv[1:]:
    for l in open(p):
        if l.startswith("{"):
            d=json.loads(l); print(p.split("/")[-1], "ms/step", round(d["ms_per_step"],3), "alone", round(d["ms_per_frame_alone"],3), "e2e", round(d["ms_per_frame_end_to_end"],3), "twice", d["frames_rendered_twice"], "GB", round(d["device_bytes"]/1e9,1), "Mrays/s", round(d["value"],1))
