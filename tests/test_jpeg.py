"""The JPEG reader (raytracer_amd/host/jpeg.cpp) against libjpeg (through PIL) on files written at test time:
baseline and progressive, every chroma subsampling, restart intervals, odd sizes, grayscale.  The two decoders use different inverse
DCTs and chroma filters, so equality is within a few counts, not exact; what the reference's own decoder would
produce is unpinned (no Rust toolchain)."""
import os

import numpy as np
import pytest

import raytracer_amd as ra

Image = pytest.importorskip("PIL.Image")


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 17.) * np.cos(y / 23.), 128 + 90 * np.cos(x / 11. + y / 29.),
                    128 + 80 * np.sin((x + y) / 13.)], -1)
    return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w", [(64, 64), (37, 51), (129, 200), (8, 8), (1, 1)])
@pytest.mark.parametrize("sub", [0, 1, 2])
@pytest.mark.parametrize("restart", [0, 3])
def test_matches_libjpeg(tmp_path, h, w, sub, restart):
    p = str(tmp_path / "t.jpg")
    kw = {"restart_marker_blocks": restart} if restart else {}
    Image.fromarray(picture(h, w, h * w + sub)).save(p, quality=88, subsampling=sub, **kw)
    ref = np.asarray(Image.open(p).convert("RGB")).astype(int)
    got = ra.read_image(p).astype(int)
    assert got.shape == ref.shape
    d = np.abs(ref - got)
    assert d.max() <= 4 and d.mean() < 0.5, (d.max(), d.mean())
    if sub == 0:
        assert d.mean() < 0.05     # no chroma filter involved: only the inverse DCTs differ


@pytest.mark.parametrize("h,w", [(64, 64), (37, 51), (129, 200), (8, 8), (1, 1)])
@pytest.mark.parametrize("sub", [0, 1, 2])
@pytest.mark.parametrize("quality,restart", [(88, 0), (97, 2), (40, 0)])
def test_progressive_matches_libjpeg(tmp_path, h, w, sub, quality, restart):
    """spectral selection + successive approximation (libjpeg's standard scan script: interleaved DC scans, one AC
    scan per component and refinement passes), with and without restart intervals"""
    p = str(tmp_path / "p.jpg")
    kw = {"restart_marker_blocks": restart} if restart else {}
    Image.fromarray(picture(h, w, h + w + sub)).save(p, quality=quality, subsampling=sub, progressive=True, **kw)
    assert Image.open(p).info.get("progressive")
    ref = np.asarray(Image.open(p).convert("RGB")).astype(int)
    got = ra.read_image(p).astype(int)
    assert got.shape == ref.shape
    d = np.abs(ref - got)
    assert d.max() <= 4 and d.mean() < 0.5, (d.max(), d.mean())
    if sub == 0:
        assert d.mean() < 0.05
    # the same picture, stored sequentially, decodes to the same samples: both paths end in the same coefficients
    q = str(tmp_path / "b.jpg")
    Image.fromarray(picture(h, w, h + w + sub)).save(q, quality=quality, subsampling=sub)
    assert np.array_equal(ra.read_image(q), ra.read_image(p))


def test_progressive_grayscale_and_truncation(tmp_path):
    g = str(tmp_path / "g.jpg")
    Image.fromarray(picture(45, 70, 4)[..., 0]).save(g, quality=90, progressive=True)
    ref = np.asarray(Image.open(g).convert("RGB")).astype(int)
    assert np.abs(ref - ra.read_image(g).astype(int)).max() <= 2
    data = open(g, "rb").read()
    cut = str(tmp_path / "cut.jpg")
    open(cut, "wb").write(data[:len(data) // 3])
    try:
        img = ra.read_image(cut)                  # whatever scans were complete, or a clean error: never a crash
        assert img.shape == (45, 70, 3)
    except RuntimeError:
        pass


def test_grayscale_png_and_bad_files(tmp_path):
    g = str(tmp_path / "g.jpg")
    Image.fromarray(picture(40, 56, 1)[..., 0]).save(g, quality=90)
    ref = np.asarray(Image.open(g).convert("RGB")).astype(int)
    assert np.abs(ref - ra.read_image(g).astype(int)).max() <= 2
    png = str(tmp_path / "x.png")
    img = picture(20, 30, 3)
    ra.write_png(png, img)
    assert np.array_equal(ra.read_image(png), img)          # told apart by content, not by name
    with pytest.raises(RuntimeError):
        ra.read_image(str(tmp_path / "missing.jpg"))
    bad = str(tmp_path / "bad.jpg")
    open(bad, "wb").write(open(g, "rb").read()[:200])
    with pytest.raises(RuntimeError):
        ra.read_image(bad)


@pytest.mark.skipif(not os.path.exists("/root/reference/assets/storforsen_4.jpg"), reason="reference checkout not present")
def test_reference_cubemap_asset_and_scenes():
    """assets/storforsen_4.jpg (8192x6144, baseline, 4:4:4) decodes like libjpeg, and the scenes that name it load"""
    a = ra.read_image("/root/reference/assets/storforsen_4.jpg")
    assert a.shape == (6144, 8192, 3)
    Image.MAX_IMAGE_PIXELS = None
    b = np.asarray(Image.open("/root/reference/assets/storforsen_4.jpg").convert("RGB"))
    d = np.abs(a[::7].astype(int) - b[::7].astype(int))
    assert d.max() <= 2 and d.mean() < 0.01
    e = ra.read_image("/root/reference/assets/earth.jpg")       # progressive, 4:2:0
    pe = np.asarray(Image.open("/root/reference/assets/earth.jpg").convert("RGB"))
    assert e.shape == pe.shape == (600, 1200, 3)
    de = np.abs(e.astype(int) - pe.astype(int))
    assert de.max() <= 4 and de.mean() < 0.05
    for name, objs in [("refraction", 4), ("fedora", 2)]:
        s = ra.SdlScene(f"/root/reference/scenes/{name}.sdl", path=True, base_dir="/root/reference")
        info = ra.desc_info(s.flatten())
        assert info["n_objects"] == objs and info["skybox_kind"] == 2
