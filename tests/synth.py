"""Synthetic inputs shared by the tests and bench.py: planar RGB uint8 images in three
value distributions -- U uniform bytes, C cell-like (bright background, dark blobs,
channels equal up to +-3 jitter: low entropy), K constant 128."""
import numpy as np

SEED0 = 0x1A2B3C4D


def rgb_image(kind, W, H, index=0):
    rng = np.random.default_rng(SEED0 ^ index)
    if kind == "U":
        return rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
    if kind == "K":
        return np.full((3, H, W), 128, np.uint8)
    if kind == "C":
        base = rng.normal(200.0, 8.0, size=(H, W))
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(12):
            cy, cx = rng.uniform(0, H), rng.uniform(0, W)
            ry, rx = rng.uniform(H / 30 + 1, H / 8 + 2), rng.uniform(W / 30 + 1, W / 8 + 2)
            inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            blob = rng.normal(90.0, 20.0, size=(H, W))
            base = np.where(inside, blob, base)
        out = np.empty((3, H, W), np.uint8)
        for c in range(3):
            jitter = rng.integers(-3, 4, size=(H, W))
            out[c] = np.clip(np.rint(base) + jitter, 0, 255).astype(np.uint8)
        return out
    raise ValueError(kind)


def rgb_batch(kind, W, H, B, first_index=0):
    return np.stack([rgb_image(kind, W, H, first_index + b) for b in range(B)])


def gray_image(kind, W, H, index=0):
    """Single-channel test image: the red plane of rgb_image, with some exact zeros so the
    skip-zero histogram rule is exercised."""
    img = rgb_image(kind, W, H, index)[0].copy()
    if kind != "K":
        rng = np.random.default_rng((SEED0 ^ index) + 77)
        img[rng.random((H, W)) < 0.02] = 0
    return img


def gray_batch(kind, W, H, B, first_index=0):
    return np.stack([gray_image(kind, W, H, first_index + b) for b in range(B)])
