"""One context over several GPUs (rt_ctx_create_multi, rt_render_frames): what the reference's single-process callers
(sdl/src/main.rs:61, :91 -> scene.rs:220) get when a node has more than one GPU.

Without a GPU: the layout of the gather (rt_multi_layout) against the host statement of the shard map, and the entry
points' argument checks.  On the one-GPU box: the context over {0} (RCCL communicator of one rank), over {0, 0} and
{0, 0, 0} (several shards on one GPU, gathered with device-to-device copies: the whole n > 1 path except the links),
each against rt_render and the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle
import raytracer_amd as ra
from raytracer_amd import _native as N
from raytracer_amd import dist as rd
from raytracer_amd import scenes
from raytracer_amd.compare import ray_counts


def _plan(vw, vh, n, frames):
    p = N.RtMultiPlan()
    rc = N.gpu().rt_multi_layout(vw, vh, n, frames, C.byref(p))
    return rc, p


@pytest.mark.parametrize("vw,vh", [(1, 1), (50, 37), (640, 360), (3840, 2160), (7680, 4320)])
@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_gather_layout_matches_the_shard_map(vw, vh, n):
    """every device sends the same count (its shard padded to shard 0's), device k's part lies at k * part_bytes of the
    gathered buffer, and the parts together hold every pixel of the frame exactly once"""
    frames = 3
    rc, p = _plan(vw, vh, n, frames)
    assert rc == N.RT_OK
    assert p.slots_per_shard == rd.max_shard_slots(vw, vh, n) == N.gpu().rt_shard_pixels(vw, vh, 0, n)
    assert all(N.gpu().rt_shard_pixels(vw, vh, k, n) <= p.slots_per_shard for k in range(n))
    assert p.part_bytes_rgb == p.slots_per_shard * 3 * frames and p.part_bytes_ids == p.slots_per_shard * 8 * frames
    assert p.gathered_bytes_rgb == n * p.part_bytes_rgb and p.gathered_bytes_ids == n * p.part_bytes_ids
    assert p.frame_bytes_rgb == vw * vh * 3
    if vw * vh <= 640 * 360:
        # fill a gathered buffer the way the devices would ([shard][frame][slot]: each slot with its pixel's index and
        # its frame's number) and put it together with the host statement of the de-tile
        g = np.zeros((n, frames, p.slots_per_shard, 4), dtype=np.uint32)
        for k in range(n):
            idx = rd.shard_pixel_indices(vw, vh, k, n)
            ok = idx >= 0
            for f in range(frames):
                g[k, f, : len(idx)][ok, 0] = idx[ok] + 1
                g[k, f, : len(idx)][ok, 1] = f + 1
        for f in range(frames):
            img = rd.detile_host(np.ascontiguousarray(g[:, f]), vw, vh, n, p.slots_per_shard, 4)
            assert np.array_equal(img[..., 0].ravel(), np.arange(vw * vh, dtype=np.uint32) + 1)
            assert (img[..., 1] == f + 1).all()


def test_layout_and_create_refuse_bad_arguments():
    assert _plan(0, 10, 2, 1)[0] == N.RT_ERR_INVALID
    assert _plan(10, 10, 0, 1)[0] == N.RT_ERR_INVALID
    assert _plan(10, 10, 2, 0)[0] == N.RT_ERR_INVALID
    assert _plan(10, 10, 2, 65)[0] == N.RT_ERR_INVALID           # RT_BATCH_MAX frames per set
    assert N.gpu().rt_multi_layout(10, 10, 2, 1, None) == N.RT_ERR_INVALID
    assert _plan(30000, 30000, 1, 4)[0] == N.RT_ERR_UNSUPPORTED   # more than 2^31 slots in one set of launches
    c = C.c_void_p()
    assert N.gpu().rt_ctx_create_multi(None, 2, C.byref(c)) == N.RT_ERR_INVALID and not c
    ids = (C.c_int * 2)(0, 1)
    assert N.gpu().rt_ctx_create_multi(ids, 0, C.byref(c)) == N.RT_ERR_INVALID and not c
    assert b"device list" in N.gpu().rt_last_error(None)
    assert N.gpu().rt_ctx_create_multi(ids, 2, None) == N.RT_ERR_INVALID
    assert N.gpu().rt_ctx_device_count(None) == 0


def test_no_device_is_an_error_not_a_fallback():
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(N.RtError) as e:
        ra.Context(devices=[0, 1])
    assert e.value.code == N.RT_ERR_DEVICE


def _path_camera(f):
    return ra.Camera(origin=(0.15 * f, 1.2 + 0.1 * f, 3.0 - 0.1 * f), yaw=0.03 * f + 1e-4, pitch=-0.2 - 0.01 * f, fov=60.)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,env", [
    ([0], {}),                                   # rt_ctx_create: no collective
    ([0], {"RTGPU_MULTI_FORCE": "1"}),           # the gather machinery with one member: ncclCommInitAll + ncclGather of one rank
    ([0, 0], {}),                                # two shards on one GPU, gathered by copies
    ([0, 0, 0], {}),                             # three: shards of unequal size
], ids=["one-device", "one-device-through-rccl", "two-shards", "three-shards"])
def test_context_over_devices_renders_the_single_device_frame(devices, env, monkeypatch, gpu_ctx):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = scenes.small_mixed_scene(203, 117, cells=12)          # ragged: edge supertiles are padded
    gpu_ctx.upload(s)
    want, want_ids, want_st = gpu_ctx.render(ids=True)
    o_rgb, o_ids, o_st, _ = oracle.render(s, 0)
    with ra.Context(devices=devices) as ctx:
        forced = env.get("RTGPU_MULTI_FORCE") == "1"
        assert ctx.n_devices == (len(devices) if len(devices) > 1 or forced else 1)
        ctx.upload(s)
        for again in range(2):                                 # the second frame is queued without level-size read-backs
            rgb, ids, st = ctx.render(ids=True)
            assert np.array_equal(rgb, want) and np.array_equal(ids, want_ids)
            assert ray_counts(st) == ray_counts(want_st) == ray_counts(o_st)
            assert list(st.rays_per_level) == list(want_st.rays_per_level)
        assert np.array_equal(rgb, o_rgb) and np.array_equal(ids, o_ids)
        rgb_only, none, _ = ctx.render()
        assert none is None and np.array_equal(rgb_only, want)
        # a camera path as one set of launches on every device, against the frames rendered one by one
        cams = [_path_camera(f) for f in range(5)]
        frames, fids, fst = ctx.render_frames(cams, ids=True)
        total = 0
        for f, cam in enumerate(cams):
            cam.vw, cam.vh = 203, 117
            s.set_camera(cam)
            gpu_ctx.upload(s)
            one, one_ids, one_st = gpu_ctx.render(ids=True)
            assert np.array_equal(frames[f], one), f
            assert np.array_equal(fids[f], one_ids), f
            total += one_st.rays
        assert fst.rays == total
        # the scene's own view held for three frames; a re-upload that keeps the meshes
        s.set_camera(cams[2])
        ctx.upload(s, keep_meshes=True)
        still, _, _ = ctx.render_frames(3)
        assert all(np.array_equal(still[f], frames[2]) for f in range(3))
        ctx.set_area_seed(5)
        assert np.array_equal(ctx.render()[0], frames[2])      # (no area lights: the seed changes nothing)


@pytest.mark.gpu
def test_context_over_devices_with_area_lights_and_errors(gpu_ctx):
    s = scenes.small_mixed_scene(96, 72, cells=6, area_lights=True)
    gpu_ctx.upload(s)
    try:
        gpu_ctx.set_area_seed(77)
        want = gpu_ctx.render()[0]
    finally:
        gpu_ctx.set_area_seed()
    with ra.Context(devices=[0, 0]) as ctx:
        with pytest.raises(N.RtError):
            ctx.render_frames(2)                               # no scene yet
        ctx.vw, ctx.vh = 96, 72
        ctx.upload(s)
        ctx.set_area_seed(77)                                  # reaches every member: shards of one frame must agree on the samples
        assert np.array_equal(ctx.render()[0], want)
        with pytest.raises(N.RtError):
            ctx.render_frames(65)
        bad = ctx.camera_array([_path_camera(0)])
        bad[0].vw = 95
        with pytest.raises(N.RtError):
            ctx.render_frames(bad)
        assert np.array_equal(ctx.render()[0], want)           # still usable after refused calls
    with pytest.raises(N.RtError):
        ra.Context(devices=[0, 99])
