"""Synthetic stereo sequences with known geometry (no KITTI data in the image).

A static cloud of point landmarks is rendered as small signed Gaussian blobs over a smooth background
into rectified left/right views from a moving camera, so FAST fires on the blob peaks, LK can track
them, stereo triangulation recovers their depth and RANSAC-PnP recovers the known motion.
Pure numpy, seeded, deterministic: the same seed gives the same bytes here and on the GPU box.
"""
import numpy as np

# calibration presets (reference calibration/*.yaml; main.cpp:357-364 for run1)
KITTI00 = dict(width=1241, height=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=-386.1448)
ZED = dict(width=1920, height=1080, fx=684.37, fy=684.37, cx=689.89, cy=406.87, bf=-82.124)
RUN1 = dict(width=512, height=288, fx=322.11376, fy=322.11376, cx=327.47336, cy=176.33722, bf=-22.5428)


def projection_matrices(cal):
    """Pl, Pr as src/stereo_vo.cpp:46-47 builds them (3x4 float32, P_r[0][3] = bf)."""
    Pl = np.array([[cal["fx"], 0, cal["cx"], 0], [0, cal["fy"], cal["cy"], 0], [0, 0, 1, 0]], np.float32)
    Pr = Pl.copy()
    Pr[0, 3] = cal["bf"]
    return Pl, Pr


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


class StereoSequence:
    """n_frames rectified stereo pairs + ground-truth camera-to-world poses.

    Scene: n_layers fronto-parallel textured planes at depths depth[0]..depth[1] (world z), each an
    infinite plane of random-gray square cells (sized to appear ~cell_px wide) with a coarse random
    see-through mask; the farthest plane is opaque.  Every pixel is rendered by exact ray/plane
    intersection from the camera pose, so stereo disparity (bf/Z) and the frame-to-frame flow are
    geometrically exact and sub-pixel accurate; cell borders are anti-aliased over ~1.5 px.
    """

    def __init__(self, cal=KITTI00, n_frames=16, seed=0x5EED0002, step=0.5, yaw_amp_deg=0.2,
                 depth=(22.0, 90.0), n_layers=6, cell_px=13.0, coverage=0.55, noise=1.0,
                 movers=0.0, mover_step=(0.6, 0.15), mover_blob=4.0):
        """movers > 0 adds an INDEPENDENTLY MOVING foreground layer covering that fraction of the pixels (blobs mover_blob texture
        cells wide), at 0.8 x the nearest depth, sliding by mover_step metres per frame in world x / y (the default is 20 px per
        frame at KITTI-00 focal length: beyond the 8 px reprojection threshold, so its tracks really are outliers).  Its features track
        perfectly well through the four LK passes (left / right views of one instant agree) but contradict the camera motion,
        i.e. they are the outliers RANSAC-PnP exists for — a static scene never makes the adaptive loop work."""
        self.cal, self.n_frames = dict(cal), n_frames
        rng = np.random.default_rng(seed)
        self.baseline = -cal["bf"] / cal["fx"]
        # layers, near -> far (geometric spacing)
        self.layer_z = np.geomspace(depth[0], depth[1], n_layers)
        self.layer_cell = self.layer_z * cell_px / cal["fx"]            # metres per cell
        self.layer_off = rng.uniform(0, 256, (n_layers, 2))
        self.tex = rng.integers(25, 231, (n_layers, 256, 256)).astype(np.float32)
        self.mask = rng.random((n_layers, 64, 64)) < coverage
        self.mask[-1] = True
        self.mask_scale = 11.0                                           # mask cells = 11 texture cells
        self.aa_px, self.cell_px, self.noise = 1.5, cell_px, noise
        self.movers, self.mover_step, self.mover_blob = float(movers), (float(mover_step[0]), float(mover_step[1])), float(mover_blob)
        if self.movers > 0:                                              # drawn after everything else: movers = 0 renders the same bytes as before
            mrng = np.random.default_rng(seed ^ 0x5A5A5A)
            self.mover_z = 0.8 * depth[0]
            self.mover_cell = self.mover_z * cell_px / cal["fx"]
            self.mover_off = mrng.uniform(0, 256, 2)
            self.mover_tex = mrng.integers(25, 231, (256, 256)).astype(np.float32)
            self.mover_mask = mrng.random((64, 64)) < self.movers
        W, H = cal["width"], cal["height"]
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        self._rx = (xx - cal["cx"]) / cal["fx"]
        self._ry = (yy - cal["cy"]) / cal["fy"]
        # camera poses: forward along +z with a small yaw sinusoid
        self.poses = []
        C = np.zeros(3)
        for k in range(n_frames):
            yaw = np.deg2rad(yaw_amp_deg) * np.sin(2 * np.pi * k / 16.0)
            Rwc = _rot_y(yaw)
            T = np.eye(4)
            T[:3, :3] = Rwc
            T[:3, 3] = C
            self.poses.append(T)
            C = C + Rwc @ np.array([0, 0, step])
        self._noise_rng = np.random.default_rng(seed ^ 0xABCDEF)
        self.left, self.right = [], []
        for k in range(n_frames):
            l, r = self._render(k)
            self.left.append(l)
            self.right.append(r)

    def _layer_value(self, l, tx, ty):
        """Anti-aliased random-cell texture of layer l at texture coords (cell units)."""
        wt = np.float32(self.aa_px / self.cell_px)
        t = tx - 0.5
        i0 = np.floor(t)
        sx = np.clip((t - i0 - 0.5) / wt + 0.5, 0, 1)
        t = ty - 0.5
        j0 = np.floor(t)
        sy = np.clip((t - j0 - 0.5) / wt + 0.5, 0, 1)
        i0 = i0.astype(np.int64) & 255
        j0 = j0.astype(np.int64) & 255
        i1, j1 = (i0 + 1) & 255, (j0 + 1) & 255
        T = self.tex[l]
        top = T[j0, i0] * (1 - sx) + T[j0, i1] * sx
        bot = T[j1, i0] * (1 - sx) + T[j1, i1] * sx
        return top * (1 - sy) + bot * sy

    def _render_view(self, Rwc, C, k=0):
        H, W = self._rx.shape
        dx = Rwc[0, 0] * self._rx + Rwc[0, 1] * self._ry + Rwc[0, 2]
        dy = Rwc[1, 0] * self._rx + Rwc[1, 1] * self._ry + Rwc[1, 2]
        dz = Rwc[2, 0] * self._rx + Rwc[2, 1] * self._ry + Rwc[2, 2]
        img = np.zeros((H, W), np.float32)
        todo = np.ones((H, W), bool)
        if self.movers > 0:                                              # the moving layer is nearest: it occludes the static scene
            sm = (self.mover_z - C[2]) / dz
            tx = ((C[0] + sm * dx - k * self.mover_step[0]) / self.mover_cell + self.mover_off[0]).astype(np.float32)
            ty = ((C[1] + sm * dy - k * self.mover_step[1]) / self.mover_cell + self.mover_off[1]).astype(np.float32)
            mi = np.floor(tx / self.mover_blob).astype(np.int64) & 63
            mj = np.floor(ty / self.mover_blob).astype(np.int64) & 63
            hit = self.mover_mask[mj, mi]
            if hit.any():
                keep_tex, keep_cell = self.tex, self.layer_cell
                self.tex = np.concatenate([self.tex, self.mover_tex[None]])     # _layer_value reads self.tex[l]
                img[hit] = self._layer_value(len(self.tex) - 1, tx[hit], ty[hit])
                self.tex, self.layer_cell = keep_tex, keep_cell
                todo &= ~hit
        for l, zl in enumerate(self.layer_z):
            s = (zl - C[2]) / dz
            tx = ((C[0] + s * dx) / self.layer_cell[l] + self.layer_off[l, 0]).astype(np.float32)
            ty = ((C[1] + s * dy) / self.layer_cell[l] + self.layer_off[l, 1]).astype(np.float32)
            mi = np.floor(tx / self.mask_scale).astype(np.int64) & 63
            mj = np.floor(ty / self.mask_scale).astype(np.int64) & 63
            hit = todo & self.mask[l][mj, mi]
            if hit.any():
                img[hit] = self._layer_value(l, tx[hit], ty[hit])
                todo &= ~hit
        img += self._noise_rng.normal(0, self.noise, img.shape).astype(np.float32)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def _render(self, k):
        T = self.poses[k]
        Rwc, C = T[:3, :3], T[:3, 3]
        Cr = C + Rwc @ np.array([self.baseline, 0, 0])
        return self._render_view(Rwc, C, k), self._render_view(Rwc, Cr, k)

    def depth_at(self, k, u, v):
        """Ground-truth depth (camera z) of left-image pixel(s) (u, v) in frame k."""
        cal = self.cal
        T = self.poses[k]
        Rwc, C = T[:3, :3], T[:3, 3]
        rx, ry = (np.asarray(u) - cal["cx"]) / cal["fx"], (np.asarray(v) - cal["cy"]) / cal["fy"]
        dx = Rwc[0, 0] * rx + Rwc[0, 1] * ry + Rwc[0, 2]
        dy = Rwc[1, 0] * rx + Rwc[1, 1] * ry + Rwc[1, 2]
        dz = Rwc[2, 0] * rx + Rwc[2, 1] * ry + Rwc[2, 2]
        out = np.zeros_like(dz)
        todo = np.ones(dz.shape, bool)
        for l, zl in enumerate(self.layer_z):
            s = (zl - C[2]) / dz
            tx = (C[0] + s * dx) / self.layer_cell[l] + self.layer_off[l, 0]
            ty = (C[1] + s * dy) / self.layer_cell[l] + self.layer_off[l, 1]
            mi = np.floor(tx / self.mask_scale).astype(np.int64) & 63
            mj = np.floor(ty / self.mask_scale).astype(np.int64) & 63
            hit = todo & self.mask[l][mj, mi]
            out[hit] = s[hit]            # camera-frame z = s (ray has unit z in the camera frame)
            todo &= ~hit
        return out

    def relative_motion(self, k):
        """Ground-truth transform the VO should return for frames (k-1 -> k): pose_{k-1}^-1 pose_k."""
        return np.linalg.inv(self.poses[k - 1]) @ self.poses[k]


def integrate(transforms, start=None):
    """frame_pose = frame_pose * T per frame (main.cpp:396); returns the list of 4x4 poses."""
    pose = np.eye(4) if start is None else start.copy()
    out = []
    for T in transforms:
        pose = pose @ T
        out.append(pose.copy())
    return out


def ate_rmse(poses_a, poses_b):
    pa = np.array([p[:3, 3] for p in poses_a])
    pb = np.array([p[:3, 3] for p in poses_b])
    return float(np.sqrt(((pa - pb) ** 2).sum(1).mean()))
