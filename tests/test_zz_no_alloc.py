"""rt_cast_rays_noalloc (probe rays staged in the library's file-scope device array, no hipMalloc) against
rt_cast_rays (per-call allocations): same bytes, on two calls in a row, the second smaller than the first."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import _native as N
from raytracer_amd import scenes

MAX = 65536   # RT_CAST_NOALLOC_MAX


def probe_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-3., 3., (n, 3)) + np.array([0., 1.5, 4.])
    d = rng.normal(size=(n, 3)) * np.array([1., 0.6, 1.]) + np.array([0., -0.3, -1.2])
    d /= np.sqrt((d * d).sum(axis=1))[:, None]
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def cast(ctx, fn, o, d):
    hits = (N.RtRayHit * len(o))()
    C.memset(hits, 0xAB, C.sizeof(hits))
    rc = fn(ctx._c, len(o), o.ctypes.data, d.ctypes.data, hits)
    assert rc == N.RT_OK, N.gpu().rt_last_error(ctx._c)
    return bytes(hits)


@pytest.mark.gpu
def test_no_alloc_probe_equals_the_allocating_one():
    L = N.gpu()
    with ra.Context(0) as ctx:
        ctx.upload(scenes.small_mixed_scene(64, 48, cells=10))
        # first call: the largest batch the entry point takes; second call: a smaller one (other rays), which must
        # not see anything the first left in the shared array
        for n, seed in ((MAX, 1), (777, 2)):
            o, d = probe_rays(n, seed)
            want = cast(ctx, L.rt_cast_rays, o, d)
            got = cast(ctx, L.rt_cast_rays_noalloc, o, d)
            assert got == want, f"n={n}"
            h = np.frombuffer(got, dtype=np.uint32).reshape(n, -1)
            assert (h[:, 0] != N.RT_ID_NONE).sum() > n // 10        # rays do hit things
        # a second context on the same GPU uses the same array
        with ra.Context(0) as other:
            other.upload(scenes.loop_scene(32, 24))
            o, d = probe_rays(300, 3)
            assert cast(other, L.rt_cast_rays_noalloc, o, d) == cast(other, L.rt_cast_rays, o, d)
        # beyond the limit: a clear error, nothing cast
        o, d = probe_rays(MAX + 1, 4)
        hits = (N.RtRayHit * (MAX + 1))()
        assert L.rt_cast_rays_noalloc(ctx._c, MAX + 1, o.ctypes.data, d.ctypes.data, hits) == N.RT_ERR_INVALID
        assert b"RT_CAST_NOALLOC_MAX" in L.rt_last_error(ctx._c)
        assert L.rt_cast_rays_noalloc(ctx._c, 0, None, None, None) == N.RT_OK
        assert L.rt_cast_rays_noalloc(ctx._c, 5, None, d.ctypes.data, hits) == N.RT_ERR_INVALID
        # and the context still works
        o, d = probe_rays(64, 5)
        assert cast(ctx, L.rt_cast_rays_noalloc, o, d) == cast(ctx, L.rt_cast_rays, o, d)
    with ra.Context(0) as fresh:
        o, d = probe_rays(8, 6)
        hits = (N.RtRayHit * 8)()
        assert L.rt_cast_rays_noalloc(fresh._c, 8, o.ctypes.data, d.ctypes.data, hits) == N.RT_ERR_INVALID   # no scene
