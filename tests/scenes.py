"""Procedural test scenes.

make_empty_image / add_triangle regenerate the synthetic scenes of the reference's own known-answer
tests (recipe: /root/reference/src/main.cpp:80-100); no reference data file is needed for them.
"""
import numpy as np


def make_empty_image(n_rows, n_cols):
    return np.zeros((n_rows, n_cols), np.uint8)


def add_triangle(image, x, y, r):
    """Right-pointing triangle of value 120 with a 1-px hole at the base centre (main.cpp:89-100)."""
    for i in range(-r, r + 1):
        for j in range(0, r + 1):
            if abs(i) <= r - abs(j):
                image[i + y, j + x] = 120
    image[y, x] = 0


def featureset_scene():
    """test_featureset (main.cpp:106-109): 300x200 (rows x cols), 11 triangles at x=20."""
    img = make_empty_image(300, 200)
    for i in range(11):
        add_triangle(img, 20, (i + 1) * 20, 8)
    return img


def circular_scene():
    """test_circularMatching (main.cpp:175-186): four 600x600 images, 11x11 triangles at 40-px pitch."""
    iL0, iR0, iL1, iR1 = (make_empty_image(600, 600) for _ in range(4))
    for i in range(11):
        for j in range(11):
            add_triangle(iL0, (j + 1) * 40, (i + 1) * 40, 8)
            add_triangle(iL1, (j + 1) * 40 + 1, (i + 1) * 40, 8)
            add_triangle(iR0, (j + 1) * 40, (i + 1) * 40 + 1, 8)
            add_triangle(iR1, (j + 1) * 40 + 1, (i + 1) * 40 + 1, 8)
    return iL0, iR0, iL1, iR1


def camera_to_world_scene():
    """test_cameraToWorld (main.cpp:211-238): 27 lattice points, K = I, 90 deg about z + (0,0,1)."""
    world, cam = [], []
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            for k in (5, 6, 7):
                world.append((-j, i, k))
                cam.append((np.float32(i) / np.float32(k + 1), np.float32(j) / np.float32(k + 1)))
    return np.eye(3, dtype=np.float32), np.array(cam, np.float32), np.array(world, np.float32)


def random_texture(h, w, seed, smooth=2):
    """Seeded band-limited random texture (u8) — trackable everywhere, used for LK / pyramid parity."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w)).astype(np.float32)
    for _ in range(smooth):
        img = (img + np.roll(img, 1, 0) + np.roll(img, -1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 1)) / 5.0
    img = (img - img.min()) / (img.max() - img.min() + 1e-9) * 255.0
    return img.astype(np.uint8)


def shift_image(img, dx, dy):
    """Integer shift with edge replication (content moves by +dx, +dy)."""
    h, w = img.shape
    ys = np.clip(np.arange(h) - dy, 0, h - 1)
    xs = np.clip(np.arange(w) - dx, 0, w - 1)
    return img[np.ix_(ys, xs)].copy()
