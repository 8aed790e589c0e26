#!/usr/bin/env python3
"""Generates the colour fixture of BASELINE.json configs[0] (`vo 400 run1`) from the reference's bundled data set.
Data only — no reference source is copied:
  * run1_bgr_{left,right}_0_47.npy.xz : stereo pairs 0..47 as the reference CLI feeds them to stereo_callback — 512x288
    8-bit BGR, interleaved (cv::imread default, main.cpp:38-46) — an lzma-compressed .npy of shape (48, 288, 512, 3);
  * run1_recorded.npz : ALL rows of run1/result.csv (the trajectory the reference recorded, 6 significant digits) and of
    run1/gt.csv.
Run in the build container only (needs /root/reference): python tests/golden/make_run1_color_fixture.py
"""
import lzma
import os

import numpy as np
from PIL import Image

REF = "/root/reference/run1"
HERE = os.path.dirname(os.path.abspath(__file__))
N = 48


def bgr(path):
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


def main():
    for cam in ("left", "right"):
        a = np.stack([bgr("%s/%s/frame%06d.png" % (REF, cam, i)) for i in range(N)])
        out = os.path.join(HERE, "run1_bgr_%s_0_%d.npy.xz" % (cam, N - 1))
        with lzma.open(out, "wb", preset=9) as f:
            np.save(f, a, allow_pickle=False)
        print(out, os.path.getsize(out), a.shape)
    res = np.loadtxt(REF + "/result.csv", delimiter=",", skiprows=1)
    gt = np.loadtxt(REF + "/gt.csv", delimiter=",", skiprows=1)
    out = os.path.join(HERE, "run1_recorded.npz")
    np.savez_compressed(out, result_csv=res, gt_csv=gt, result_header="x,y,z,gtx,gty", gt_header="time,x,y,dx,dy")
    print(out, os.path.getsize(out), res.shape, gt.shape)


if __name__ == "__main__":
    main()
