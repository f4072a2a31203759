"""Seeded synthetic inputs: there is no dataset and no network, so the tests and bench.py
generate EuRoC-shaped images here.  Pure numpy; deterministic for a given seed."""
import numpy as np


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_image(w=752, h=480, seed=0, n_shapes=None, noise_sigma=2.0):
    """u8 HxW image: 3 octaves of value noise + random rectangles/discs + Gaussian pixel noise."""
    rng = np.random.default_rng(0x4D0B + seed)
    img = 96 * _value_noise(rng, h, w, 64) + 64 * _value_noise(rng, h, w, 16) + 32 * _value_noise(rng, h, w, 4)
    img += 20
    if n_shapes is None:
        n_shapes = int(400 * (w * h) / 360960)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_shapes):
        cx, cy = rng.integers(0, w), rng.integers(0, h)
        s = int(rng.integers(3, 40))
        grey = float(rng.integers(0, 256))
        kind = rng.integers(0, 3)
        x0, x1 = max(cx - s, 0), min(cx + s + 1, w)
        y0, y1 = max(cy - s, 0), min(cy + s + 1, h)
        if x1 <= x0 or y1 <= y0:
            continue
        sub = img[y0:y1, x0:x1]
        if kind == 0:
            sub[:] = grey
        elif kind == 1:
            m = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= s * s
            sub[m] = grey
        else:  # rotated rectangle
            th = rng.random() * np.pi
            u = (xx[y0:y1, x0:x1] - cx) * np.cos(th) + (yy[y0:y1, x0:x1] - cy) * np.sin(th)
            v = -(xx[y0:y1, x0:x1] - cx) * np.sin(th) + (yy[y0:y1, x0:x1] - cy) * np.cos(th)
            m = (np.abs(u) <= s * 0.8) & (np.abs(v) <= s * 0.4)
            sub[m] = grey
    img += rng.normal(0, noise_sigma, size=img.shape)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def make_stereo_pair(w=752, h=480, seed=0, dmin=2.0, dmax=60.0):
    """(left, right): right = left warped by a smooth horizontal disparity field d(x,y) in [dmin, dmax]
    (left pixel (x,y) appears at (x-d, y) in the right image), plus independent pixel noise."""
    left = make_image(w, h, seed)
    rng = np.random.default_rng(0x57E0 + seed)
    d = dmin + (dmax - dmin) * _value_noise(rng, h, w, 128)
    xs = np.arange(w, dtype=np.float32)[None, :] + d  # right(x) samples left(x + d)
    x0 = np.clip(np.floor(xs).astype(np.int32), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    f = xs - np.floor(xs)
    rows = np.arange(h)[:, None]
    L = left.astype(np.float32)
    right = L[rows, x0] * (1 - f) + L[rows, x1] * f + rng.normal(0, 1.0, size=L.shape)
    return left, np.ascontiguousarray(np.clip(np.rint(right), 0, 255).astype(np.uint8))


def shift_image(img, dx, dy):
    """Integer global shift with edge replication (feeds the frame-to-frame matchers)."""
    h, w = img.shape
    ys = np.clip(np.arange(h) - dy, 0, h - 1)
    xs = np.clip(np.arange(w) - dx, 0, w - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def make_vocabulary(k=10, L=6, seed=0):
    """Synthetic DBoW2-shaped ORB vocabulary (the real ORBvoc.txt is a missing blob, SURVEY.md finding 3):
    complete k-ary tree of depth L in level order (node 0 = root, children of n = k*n+1 .. k*n+k), each child
    descriptor = parent descriptor with a level-dependent fraction of random bits flipped.
    Returns (nodeDesc [nnodes, 32] u8, firstChild [nnodes] i32, -1 for leaves)."""
    rng = np.random.default_rng(0xB0 + seed)
    nn = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((nn, 32), np.uint8)
    first = np.full(nn, -1, np.int32)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    start, cnt = 0, 1
    for lvl in range(L):
        parents = np.arange(start, start + cnt)
        first[parents] = k * parents + 1
        p = 0.5 / (lvl + 1.5)
        child_ids = (k * parents[:, None] + 1 + np.arange(k)[None, :]).reshape(-1)
        flips = np.packbits(rng.random((len(child_ids), 256)) < p, axis=1)
        desc[child_ids] = np.repeat(desc[parents], k, axis=0) ^ flips
        start, cnt = start + cnt, cnt * k
    return desc, first
