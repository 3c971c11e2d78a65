"""Image / id-buffer comparison metrics used by the parity tests and the bench report."""
import numpy as np


def image_metrics(a, b):
    """a, b: uint8 arrays of equal shape (..., 3) -> dict(exact, within1, max_abs, psnr)"""
    a = np.asarray(a).reshape(-1, 3).astype(np.int32)
    b = np.asarray(b).reshape(-1, 3).astype(np.int32)
    d = np.abs(a - b)
    per_px = d.max(axis=1)
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    return {"exact": float((per_px == 0).mean()), "within1": float((per_px <= 1).mean()),
            "max_abs": int(d.max()) if d.size else 0, "psnr": psnr, "n": int(len(per_px))}


def id_match(a, b):
    a = np.asarray(a).reshape(-1, 2)
    b = np.asarray(b).reshape(-1, 2)
    return float((a == b).all(axis=1).mean())


def ray_counts(st):
    return (int(st.rays_primary), int(st.rays_reflect), int(st.rays_transmit), int(st.rays_shadow))
