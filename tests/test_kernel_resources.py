"""The resource figures that decide how many waves of the hot kernels a CU holds, read from the built library (no GPU):
the gfx950 code objects inside raytracer_amd/librtgpu.so carry per-kernel metadata (LDS bytes, VGPRs, scratch).

A gfx950 CU hands out its 160 KB of LDS in granules of 1,280 bytes (128 of them) and its 512 VGPRs per SIMD lane in
blocks of 8.  The shadow packet kernel is tuned to seven workgroups of four waves per CU: that needs at most 18 granules
(23,040 bytes) and at most 72 VGPRs -- with 48 bytes more of LDS a workgroup takes a nineteenth granule and only six fit,
which cost 8 % of the kernel's time before the pair queue was trimmed.  This test fails when a change to the kernels
crosses one of those lines."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "raytracer_amd", "librtgpu.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
LDS_GRANULE, LDS_GRANULES = 1280, 128


def kernel_metadata(tmp_path):
    """{mangled kernel name: {'lds': bytes, 'vgpr': n, 'scratch': bytes}} of every gfx950 kernel in the library"""
    fat = str(tmp_path / "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos, k = {}, 0, 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s:
                co = str(tmp_path / f"co{k}.elf")
                k += 1
                open(co, "wb").write(blob[i + o:i + o + s])
                notes = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
                cur = {}
                for line in notes.splitlines():
                    m = re.match(r"\s*-?\s*\.(group_segment_fixed_size|private_segment_fixed_size|vgpr_count|name):\s*(\S+)", line)
                    if not m:
                        continue
                    key, val = m.groups()
                    if key == "name":
                        cur["name"] = val
                    else:
                        cur[{"group_segment_fixed_size": "lds", "private_segment_fixed_size": "scratch", "vgpr_count": "vgpr"}[key]] = int(val)
                    if len(cur) == 4:
                        out[cur.pop("name")] = cur
                        cur = {}
        pos = i + 24
    return out


def waves_per_simd(m, waves_per_group=4):
    by_lds = LDS_GRANULES // -(-m["lds"] // LDS_GRANULE) if m["lds"] else 99
    by_vgpr = 512 // (-(-m["vgpr"] // 8) * 8)
    return min(8, by_vgpr, by_lds * waves_per_group // 4)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF) and shutil.which("objcopy")), reason="needs the built library, llvm-readelf and objcopy")
def test_hot_kernels_keep_their_waves_per_simd(tmp_path):
    md = kernel_metadata(tmp_path)
    assert len(md) > 30
    shadow = {k: v for k, v in md.items() if k.startswith("_Z8k_shadowI")}
    assert len(shadow) == 6                                   # COUNT x BUDGET, and the two variants without 1/direction
    for name, m in shadow.items():
        assert m["lds"] <= 18 * LDS_GRANULE, (name, m)        # a seventh workgroup per CU
        assert waves_per_simd(m) == 7, (name, m)
    # the variants config 4 runs (not counted; rays without 1/direction): no scratch memory at all
    lean_shadow = md["_Z8k_shadowILb0ELb0ELb0EEvPK6DScenePKjPKdPhjP9DCountersP12DLevelCountsPj10LongParams"]
    assert lean_shadow["scratch"] == 0
    lean_trace = [v for k, v in md.items() if k.startswith("_Z16k_trace_deferredILb0ELb0EE")]
    assert len(lean_trace) == 1 and lean_trace[0]["scratch"] == 0 and waves_per_simd(lean_trace[0]) == 5
    for k, v in md.items():
        if k.startswith("_Z16k_trace_deferredI") or k.startswith("_Z14k_trace_packetI"):
            assert waves_per_simd(v) >= 5, (k, v)
    light = md["_Z7k_lightPK6DScene6DLeveljPKyPKhj"]
    assert light["scratch"] == 0 and waves_per_simd(light) == 8
    spawn = [v for k, v in md.items() if k.startswith("_Z7k_spawnILb0ELb0EE")]
    assert len(spawn) == 1 and spawn[0]["scratch"] == 0 and waves_per_simd(spawn[0]) >= 6
    # the kernels of scenes with area lights are kernels of their own (the others are what they were without them);
    # they keep the shadow packets' waves, and nothing of theirs spills
    area = {k: v for k, v in md.items() if k.startswith("_Z13k_shadow_areaILb0E") or k.startswith("_Z12k_light_area")}
    assert len(area) == 3, sorted(area)
    for k, v in area.items():
        # (the variant with a step budget, for small launches only, keeps three values in scratch memory)
        assert v["scratch"] <= (16 if "areaILb0ELb1E" in k else 0), (k, v)
        if "shadow" in k:
            assert v["lds"] <= 18 * 1280 and waves_per_simd(v) == 7, (k, v)
